"""Sharding of the block-transform path over the GPUs of one node -- no PyTorch.

Blocks -- and therefore planes and block rows -- are independent units (no DC prediction, no
inter-block state anywhere in steps 4-6; the reference's pipeline/__init__.py:104-106 processes
bands one after another with nothing shared), so the data path needs NO collective: every rank
transforms its own contiguous range.  The only exchange is the final gather of the int16
coefficient stream to one rank over RCCL/xGMI.

Pieces (one process per GPU):
  * ``shard_*``          contiguous unit ranges per rank;
  * ``launch_ranks``     start N copies of a script with the RANK / LOCAL_RANK / WORLD_SIZE /
                         MASTER_* environment ``torch.distributed.run`` would give them (so a
                         script works the same under either launcher);
  * ``ControlPlane``     the small-message side channel between the ranks (TCP on 127.0.0.1:
                         barrier, all-gather / max / min of scalars, broadcast of the 128-byte
                         RCCL id).  Bulk data never goes through it;
  * ``NativeComm``       the RCCL communicator owned by libjpegx (``jpegx_comm_*``);
  * ``transform_and_gather``  the multi-GPU driver step: forward transform of the rank's
                         planes chunk by chunk on one HIP stream while a second stream ships every
                         finished chunk to the root (grouped ncclSend/ncclRecv), i.e. the gather
                         is overlapped with the compute.
"""
import base64
import json
import os
import socket
import struct
import subprocess
import sys
import time


# ---------------------------------------------------------------------------------------------
# shard planning
# ---------------------------------------------------------------------------------------------
def shard_range(n_units, world, rank):
    """Contiguous [lo, hi) of ``n_units`` for ``rank``: sizes differ by at most one, low ranks first."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank %r / world %r" % (rank, world))
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_planes(n_planes, world, rank):
    """Plane ids owned by ``rank`` when a batch of independent planes is split over the ranks."""
    return shard_range(n_planes, world, rank)


def shard_block_rows(height, world, rank):
    """Row range [y0, y1) (multiples of 8) of ONE tall plane owned by ``rank``: contiguous block rows,
    so that both the rank's input rows and its slice of the zigzag stream are single spans."""
    if height % 8:
        raise ValueError("plane height must be a multiple of 8")
    lo, hi = shard_range(height // 8, world, rank)
    return lo * 8, hi * 8


def chunk_spans(n_units, chunk):
    """[lo, hi) spans of at most ``chunk`` units covering 0..n_units."""
    chunk = max(1, int(chunk))
    return [(lo, min(n_units, lo + chunk)) for lo in range(0, n_units, chunk)]


# ---------------------------------------------------------------------------------------------
# process launcher
# ---------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_process_env(env=None):
    """Environment every rank process needs BEFORE anything loads HIP / RCCL, whoever launched it (set with
    setdefault: an explicit choice of the caller wins).

    * ``HSA_ENABLE_IPC_MODE_LEGACY=0``: the host driver only supports dmabuf IPC; without it RCCL's peer-to-peer
      setup fails in hipIpcGetMemHandle.
    * ``RSMI_MUTEX_THREAD_ONLY=1``: RCCL's communicator creation queries rocm_smi, which serialises its device
      queries through a PROCESS-SHARED pthread mutex kept in ``/dev/shm/rocm_smi_*``.  A process that ended
      uncleanly while holding it (seen after rocprofv3 passes on the same box, round 2) leaves it locked: the next
      rsmi_init times out on it after 5 s ("init_mutex ... unlock timed lock, ret: 1") and the blocking lock of the
      first device query behind it never returns -- both ranks "sat in communicator creation".  With this variable
      rocm_smi uses a process-local mutex instead, which is all one-process-per-GPU ranks need (they only read sysfs).
    * ``JPEGX_COMM_LOG=1``: libjpegx logs the phases of the RCCL calls to stderr with time stamps.
    """
    env = os.environ if env is None else env
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("RSMI_MUTEX_THREAD_ONLY", "1")
    env.setdefault("JPEGX_COMM_LOG", "1")
    return env


def rsmi_shm_report(shm_dir="/dev/shm"):
    """State of rocm_smi's process-shared mutexes (``/dev/shm/rocm_smi_*``, a pthread_mutex_t at offset 0): for each
    file the lock word, the owner's thread id and whether that thread still exists.  ``stale`` = locked by a thread
    that is gone: the condition behind round 2's hang in communicator creation (see rank_process_env)."""
    out = []
    try:
        names = sorted(n for n in os.listdir(shm_dir) if n.startswith("rocm_smi"))
    except OSError:
        return out
    for n in names:
        rec = {"file": n}
        try:
            with open(os.path.join(shm_dir, n), "rb") as f:
                head = f.read(16)
            if len(head) >= 12:
                lock, _count, owner = struct.unpack("<IIi", head[:12])
                rec.update(lock=lock, owner_tid=owner)
                tid = owner if owner > 0 else (lock & 0x3FFFFFFF)
                alive = bool(tid) and os.path.exists("/proc/%d" % tid)
                rec["owner_alive"] = alive
                rec["stale"] = bool(lock != 0 and tid and not alive)
        except OSError as exc:
            rec["error"] = "%s: %s" % (type(exc).__name__, exc)
        out.append(rec)
    return out


def launch_ranks(nranks, argv, extra_env=None, poll_s=0.05, grace_s=10.0):
    """Run ``python argv...`` once per rank and return the job's exit code (0 iff every rank
    returned 0).  Must be called before the calling process has touched the GPU.  When a rank
    fails the others are given ``grace_s`` seconds and are then terminated by PID."""
    port = _free_port()
    token = base64.b16encode(os.urandom(16)).decode()      # what a rank must present to join this job's control plane
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env["JPEGX_CTL_TOKEN"] = token
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks),
                    "LOCAL_WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "JPEGX_LAUNCHER": "jpegx.multigpu.launch_ranks"})
        rank_process_env(env)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rc, failed_at = 0, None
    while True:
        alive = 0
        for p in procs:
            code = p.poll()
            if code is None:
                alive += 1
            elif code != 0 and rc == 0:
                rc, failed_at = code, time.monotonic()
        if alive == 0:
            return rc
        if failed_at is not None and time.monotonic() - failed_at > grace_s:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc
        time.sleep(poll_s)


def rank_env():
    """(rank, local_rank, world) from the launcher's environment (1 process: 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


# ---------------------------------------------------------------------------------------------
# control plane
# ---------------------------------------------------------------------------------------------
class ControlPlaneError(RuntimeError):
    pass


class ControlPlane:
    """Star of TCP connections on 127.0.0.1 with rank 0 in the centre; JSON messages.

    Rendezvous: rank 0 listens on an ephemeral port and publishes it in a file named after
    MASTER_PORT (``torch.distributed.run`` keeps MASTER_PORT itself busy with its own store, so it
    cannot be bound here); the other ranks poll the file, connect and say who they are.  A stale
    file (an earlier job on the same MASTER_PORT) points at a dead or foreign port: the hello is
    refused and the rank polls again.  Every socket operation has a timeout, so a rank that died
    turns into an exception on the others instead of a hang.
    """
    MAGIC = "jpegx-ctl-1"

    def __init__(self, rank=None, world=None, key=None, timeout=900.0, connect_timeout=180.0):
        env_rank, _, env_world = rank_env()
        self.rank = env_rank if rank is None else int(rank)
        self.world = env_world if world is None else int(world)
        self.timeout = float(timeout)
        self._peers = {}          # rank 0: rank -> socket
        self._up = None           # others: socket to rank 0
        self._path = None
        if self.world <= 1:
            return
        if key is None:
            key = "%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
        self.key = "".join(c if c.isalnum() or c in "._-" else "_" for c in str(key))
        self._token = os.environ.get("JPEGX_CTL_TOKEN") or None     # launch_ranks hands one over; else rank 0 makes one
        self._path = os.path.join(self._rendezvous_dir(), "jpegx_ctl_%d_%s" % (os.getuid(), self.key))
        if self.rank == 0:
            self._serve(connect_timeout)
        else:
            self._join(connect_timeout)

    @staticmethod
    def _rendezvous_dir():
        """A directory only this user can write to: JPEGX_CTL_DIR if given, else XDG_RUNTIME_DIR, else a 0700
        directory of our own under /tmp (never /tmp itself: there another user could plant the file or a symlink)."""
        base = os.environ.get("JPEGX_CTL_DIR") or os.environ.get("XDG_RUNTIME_DIR")
        if base and os.path.isdir(base):
            st = os.stat(base)
            if st.st_uid == os.getuid() and os.access(base, os.W_OK):
                return base
        own = os.path.join("/tmp", "jpegx_ctl_%d.d" % os.getuid())
        try:
            os.mkdir(own, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(own)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise ControlPlaneError("%s exists but is not a private directory of uid %d" % (own, os.getuid()))
        return own

    # -- wire format: 4-byte big-endian length + UTF-8 JSON -----------------------------------
    @staticmethod
    def _send(sock, obj):
        data = json.dumps(obj).encode()
        sock.sendall(struct.pack(">I", len(data)) + data)

    @staticmethod
    def _recv(sock):
        def exactly(n):
            buf = b""
            while len(buf) < n:
                part = sock.recv(n - len(buf))
                if not part:
                    raise ControlPlaneError("control-plane peer closed the connection")
                buf += part
            return buf
        (n,) = struct.unpack(">I", exactly(4))
        return json.loads(exactly(n).decode())

    def _serve(self, connect_timeout):
        srv = socket.socket()
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(("127.0.0.1", 0))
        srv.listen(self.world)
        port = srv.getsockname()[1]
        if self._token is None:
            self._token = base64.b16encode(os.urandom(16)).decode()
        tmp = "%s.%d.tmp" % (self._path, os.getpid())
        try:
            os.unlink(tmp)
        except OSError:
            pass
        # O_EXCL | O_NOFOLLOW: never write through something another process put there; 0600: the token in it is
        # what a joining rank must echo, so only this user's processes can join
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as f:
            f.write("%d %d %s\n" % (port, os.getpid(), self._token))
        os.replace(tmp, self._path)                        # atomic publish (replaces a stale file)
        deadline = time.monotonic() + connect_timeout
        try:
            while len(self._peers) < self.world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise ControlPlaneError("only %d of %d ranks joined the control plane within %.0f s"
                                            % (len(self._peers) + 1, self.world, connect_timeout))
                conn.settimeout(10.0)
                try:
                    hello = self._recv(conn)
                except Exception:
                    conn.close()
                    continue
                ok = (isinstance(hello, dict) and hello.get("magic") == self.MAGIC and hello.get("key") == self.key
                      and hello.get("token") == self._token
                      and hello.get("world") == self.world and isinstance(hello.get("rank"), int)
                      and 0 < hello["rank"] < self.world and hello["rank"] not in self._peers)
                self._send(conn, {"ok": bool(ok)})
                if not ok:
                    conn.close()
                    continue
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[hello["rank"]] = conn
        finally:
            srv.close()

    def _join(self, connect_timeout):
        deadline = time.monotonic() + connect_timeout
        last = "no rendezvous file %s" % self._path
        while time.monotonic() < deadline:
            try:
                with open(self._path) as f:
                    fields = f.read().split()
                port = int(fields[0])
                token = fields[2] if len(fields) > 2 else ""
                if self._token is not None and token != self._token:
                    raise ValueError("rendezvous file of another job (token mismatch)")
                s = socket.create_connection(("127.0.0.1", port), timeout=5.0)
                try:
                    s.settimeout(10.0)
                    self._send(s, {"magic": self.MAGIC, "key": self.key, "world": self.world, "rank": self.rank, "token": token})
                    if self._recv(s).get("ok"):
                        s.settimeout(self.timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._up = s
                        return
                    last = "hello refused by the process on port %d" % port
                except Exception as exc:
                    last = "%s: %s" % (type(exc).__name__, exc)
                s.close()
            except (OSError, ValueError, IndexError) as exc:
                last = "%s: %s" % (type(exc).__name__, exc)
            time.sleep(0.05)
        raise ControlPlaneError("rank %d could not join the control plane: %s" % (self.rank, last))

    # -- collectives on small JSON-serialisable values ---------------------------------------
    def allgather(self, value):
        """Every rank's ``value`` as a list in rank order, on every rank."""
        if self.world <= 1:
            return [value]
        try:
            if self.rank == 0:
                vals = [value] + [self._recv(self._peers[r]) for r in range(1, self.world)]
                for r in range(1, self.world):
                    self._send(self._peers[r], vals)
                return vals
            self._send(self._up, value)
            return self._recv(self._up)
        except (OSError, ValueError) as exc:
            raise ControlPlaneError("control-plane exchange failed on rank %d: %s: %s"
                                    % (self.rank, type(exc).__name__, exc))

    def barrier(self):
        self.allgather(None)

    def allreduce_max(self, x):
        return max(self.allgather(x))

    def allreduce_min(self, x):
        return min(self.allgather(x))

    def all_ok(self, ok):
        """True iff ``ok`` on every rank: taken BEFORE entering an RCCL call so that a rank-local
        failure makes every rank skip the collective instead of leaving the others blocked in it."""
        return all(bool(v) for v in self.allgather(bool(ok)))

    def bcast_bytes(self, data, src=0):
        enc = base64.b64encode(data).decode() if self.rank == src and data is not None else None
        return base64.b64decode(self.allgather(enc)[src])

    def close(self):
        for s in list(self._peers.values()) + ([self._up] if self._up else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._up = {}, None
        if self.rank == 0 and self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---------------------------------------------------------------------------------------------
# RCCL communicator (libjpegx jpegx_comm_*)
# ---------------------------------------------------------------------------------------------
class NativeComm:
    """RCCL communicator owned by libjpegx on the calling thread's current device.

    ``exchange_id`` takes rank 0's 128-byte id (bytes on rank 0, None elsewhere) and returns it on
    every rank -- ``ControlPlane.bcast_bytes`` here; an MPI bcast or a shared file would do too.
    """

    @staticmethod
    def unique_id():
        """ncclGetUniqueId through libjpegx: the 128 bytes rank 0 hands to every rank."""
        import ctypes
        import jpegx
        buf = ctypes.create_string_buffer(128)
        jpegx.check(jpegx.lib().jpegx_comm_unique_id(buf), "jpegx_comm_unique_id")
        return buf.raw

    def __init__(self, nranks, rank, exchange_id=None, timeout_s=0.0, ident=None):
        """``timeout_s``: deadline of communicator creation and of every later enqueue wait (0: JPEGX_COMM_TIMEOUT_S
        from the environment, else 120 s); a JpegxError carrying libjpegx's JPEGX_E_TIMEOUT text when it runs out.
        ``ident``: the 128-byte id when the caller has already distributed it (then ``exchange_id`` is not used)."""
        import ctypes
        import jpegx
        self._jpegx = jpegx
        L = jpegx.lib()
        t0 = time.monotonic()
        log = os.environ.get("JPEGX_COMM_LOG", "0") not in ("", "0")

        def say(what):
            if log:
                sys.stderr.write("[jpegx comm rank %d/%d pid %d +%.3fs] %s\n" % (rank, nranks, os.getpid(), time.monotonic() - t0, what))
                sys.stderr.flush()
        if ident is None:
            if rank == 0:
                ident = self.unique_id()
            say("id-broadcast enter")
            ident = exchange_id(ident)
            say("id-broadcast done")
        stale = [r for r in rsmi_shm_report() if r.get("stale")]
        if stale:
            say("WARNING stale rocm_smi shared mutex (locked by a thread that is gone): %s; RSMI_MUTEX_THREAD_ONLY=%s"
                % (json.dumps(stale), os.environ.get("RSMI_MUTEX_THREAD_ONLY", "unset")))
        handle = ctypes.c_void_p()
        jpegx.check(L.jpegx_comm_create_deadline(ctypes.byref(handle), nranks, rank, ident, float(timeout_s)), "jpegx_comm_create")
        self.handle, self.nranks, self.rank = handle.value, nranks, rank

    def count(self):
        """Number of ranks RCCL itself reports for this communicator (ncclCommCount)."""
        import ctypes
        n = ctypes.c_int(0)
        self._jpegx.check(self._jpegx.lib().jpegx_comm_count(self.handle, ctypes.byref(n)), "jpegx_comm_count")
        return n.value

    def gather_bytes(self, send_ptr, send_bytes, recv_ptr=None, recv_bytes=None, recv_offsets=None, root=0, stream=None):
        """Every rank sends ``send_bytes`` from ``send_ptr`` (0 = nothing to send); the root receives
        ``recv_bytes[r]`` bytes from rank r at ``recv_ptr + recv_offsets[r]`` (default: running
        offsets).  Enqueues on ``stream``; nothing is synchronised."""
        import ctypes
        n = self.nranks
        sizes = (ctypes.c_size_t * n)(*([0] * n))
        offs = (ctypes.c_size_t * n)(*([0] * n))
        if self.rank == root:
            run = 0
            for r in range(n):
                sizes[r] = int(recv_bytes[r])
                offs[r] = int(recv_offsets[r]) if recv_offsets is not None else run
                run += int(recv_bytes[r])
        self._jpegx.check(self._jpegx.lib().jpegx_comm_gather_bytes(self.handle, send_ptr, int(send_bytes), recv_ptr, sizes, offs,
                                                                      root, stream), "jpegx_comm_gather_bytes")

    def close(self):
        if self.handle:
            self._jpegx.lib().jpegx_comm_destroy(self.handle)
            self.handle = None

    def abort(self):
        """ncclCommAbort: drop outstanding operations (after a failed or timed-out round) and free the handle."""
        if self.handle:
            self._jpegx.lib().jpegx_comm_abort(self.handle)
            self.handle = None


# ---------------------------------------------------------------------------------------------
# the multi-GPU driver step
# ---------------------------------------------------------------------------------------------
class GatherPlan:
    """Where every rank's planes sit in the root's stream: plane p of the batch at byte
    ``p * plane_stream_bytes`` -- concatenating the ranks in order IS the un-sharded stream."""

    def __init__(self, n_planes, world, plane_stream_bytes, chunk_planes):
        self.world, self.plane_bytes = int(world), int(plane_stream_bytes)
        self.spans = [shard_planes(n_planes, world, r) for r in range(world)]
        self.chunk = max(1, int(chunk_planes))
        self.rounds = max((hi - lo + self.chunk - 1) // self.chunk for lo, hi in self.spans) if n_planes else 0

    def round_of(self, rank, k):
        """(first plane, count) that ``rank`` ships in round k (count 0 when it has run out)."""
        lo, hi = self.spans[rank]
        a = min(hi, lo + k * self.chunk)
        return a, min(hi, a + self.chunk) - a


def ship_round(comm, plan, k, stream_ptr, root_ptr, comm_stream, root=0, loopback=False):
    """Round k of the gather: this rank's k-th chunk of planes (its int16 stream starts at ``stream_ptr``) goes to
    the root, which posts the matching receives at the planes' offsets of ``root_ptr`` -- one grouped
    ncclSend/ncclRecv on ``comm_stream``.  The root's own planes are already in place (unless ``loopback``)."""
    rank = comm.rank
    first, count = plan.round_of(rank, k)
    out_plane = plan.plane_bytes
    sizes = [0] * plan.world
    offs = [0] * plan.world
    if rank == root:
        for r in range(plan.world):
            if r == root and not loopback:
                continue
            f, c = plan.round_of(r, k)
            sizes[r], offs[r] = c * out_plane, f * out_plane
    send = 0 if (rank == root and not loopback) else count * out_plane
    comm.gather_bytes(stream_ptr + (first - plan.spans[rank][0]) * out_plane, send, root_ptr, sizes, offs, root=root,
                      stream=comm_stream)


def transform_and_gather(comm, plan, in_ptr, stream_ptr, root_ptr, size, mode, param, flags, compute_stream, comm_stream,
                         events, root=0, loopback=False):
    """Forward-transform the rank's planes (``size`` x ``size`` fp32 each, stacked at ``in_ptr``) in
    chunks of ``plan.chunk`` planes on ``compute_stream``; after each chunk an event releases
    ``comm_stream``, which ships the chunk's int16 stream to the root with one grouped
    ncclSend/ncclRecv round (``comm.gather_bytes``).  The root's own planes are written straight
    into its slot of the gathered stream (``root_ptr``), so nothing is copied for them.
    ``stream_ptr``: where this rank's stream starts (on the root: root_ptr + its plane offset).
    ``loopback`` (single-GPU rehearsal of the control flow with a 1-rank communicator): the rank
    also ships its own chunks to itself through RCCL, into ``root_ptr``, which must then be a
    buffer distinct from ``stream_ptr``.  Enqueue-only; the caller synchronises both streams."""
    import jpegx
    L = jpegx.lib()
    rank = comm.rank
    lo_mine = plan.spans[rank][0]
    in_plane, out_plane = size * size * 4, plan.plane_bytes
    for k in range(plan.rounds):
        first, count = plan.round_of(rank, k)
        if count:
            off = first - lo_mine
            jpegx.forward_fused_device(in_ptr + off * in_plane, size * count, size, stream_ptr + off * out_plane,
                                       mode, param, flags, stream=compute_stream)
        jpegx.check(L.jpegx_event_record(events[k].handle, compute_stream), "jpegx_event_record")
        jpegx.check(L.jpegx_stream_wait_event(comm_stream, events[k].handle), "jpegx_stream_wait_event")
        ship_round(comm, plan, k, stream_ptr, root_ptr, comm_stream, root=root, loopback=loopback)
