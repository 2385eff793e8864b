"""Multi-GPU path on the CPU (no GPU here): shard planning, the rank launcher, the TCP control
plane and bench.py's N > 1 control flow (``--dry-run``) -- started bare (own launcher) and under
``torch.distributed.run`` the way the driver starts it.  A world_size-2 gloo run checks that
contiguous plane / block-row shards plus a rank-ordered gather reproduce the un-sharded
coefficient stream exactly (the per-rank transform is stood in for by the oracle)."""
import json
import os
import socket
import subprocess
import time
import sys

import pytest

from conftest import PKG, REPO


def test_shard_ranges_cover_everything_once():
    from jpegx.multigpu import shard_block_rows, shard_planes, shard_range
    for n in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert shard_planes(1024, 8, 3) == (384, 512)
    assert shard_block_rows(4096, 8, 7) == (3584, 4096)
    with pytest.raises(ValueError):
        shard_block_rows(100, 2, 0)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_gather_plan_places_every_plane_exactly_once():
    """Every plane of the batch is shipped in exactly one round by exactly its owner, at its own
    offset of the root's stream; ranks that run out early send nothing (uneven shards)."""
    from jpegx.multigpu import GatherPlan
    for n, world, chunk in ((1024, 8, 4), (10, 3, 4), (5, 2, 1), (7, 8, 2), (16, 1, 3)):
        plan = GatherPlan(n, world, 100, chunk)
        seen = []
        for k in range(plan.rounds):
            for r in range(world):
                first, count = plan.round_of(r, k)
                assert 0 <= count <= chunk
                lo, hi = plan.spans[r]
                assert count == 0 or (lo <= first and first + count <= hi)
                seen.extend(range(first, first + count))
        assert sorted(seen) == list(range(n))
        for r in range(world):
            assert plan.round_of(r, plan.rounds)[1] == 0


@pytest.mark.parametrize("world, n_planes, chunk", [(1, 3, 2), (2, 7, 4), (3, 7, 2), (4, 10, 1), (8, 1024, 4), (8, 13, 5), (5, 5, 9)])
def test_driver_step_matches_sends_and_receives_for_every_world(monkeypatch, world, n_planes, chunk):
    """jpegx.multigpu.transform_and_gather played rank by rank on the CPU with the device calls replaced by recorders:
    every rank transforms each of its planes exactly once, into its own slot; round by round what a rank sends is
    what the root has posted a receive for (same byte count, landing at that plane's offset of the root's stream);
    the root sends nothing and receives nothing from itself; event and wait calls pair up round by round."""
    import jpegx
    from jpegx import multigpu
    plane_in, plane_out, size = 64 * 64 * 4, 64 * 64 * 2, 64
    plan = multigpu.GatherPlan(n_planes, world, plane_out, chunk)

    class FakeLib:
        def __init__(self):
            self.log = []

        def jpegx_event_record(self, ev, stream):
            self.log.append(("record", ev, stream))
            return 0

        def jpegx_stream_wait_event(self, stream, ev):
            self.log.append(("wait", ev, stream))
            return 0

    class Ev:
        def __init__(self, k):
            self.handle = 1000 + k

    class Recorder:
        def __init__(self, rank):
            self.rank, self.nranks, self.calls = rank, world, []

        def gather_bytes(self, send_ptr, send_bytes, recv_ptr, recv_bytes, recv_offsets, root=0, stream=None):
            self.calls.append((send_ptr, send_bytes, recv_ptr, list(recv_bytes), list(recv_offsets), root, stream))

    per_rank = {}
    for rank in range(world):
        lo, hi = plan.spans[rank]
        fake, launches = FakeLib(), []
        monkeypatch.setattr(jpegx, "lib", lambda fake=fake: fake)
        monkeypatch.setattr(jpegx, "check", lambda rc, what="": None)
        monkeypatch.setattr(jpegx, "forward_fused_device",
                            lambda in_ptr, h, w, out_ptr, mode, param, flags, stream=None, launches=launches: launches.append((in_ptr, h, w, out_ptr, stream)))
        in_base, root_base = 1 << 40, 1 << 44
        stream_ptr = (root_base + lo * plane_out) if rank == 0 else (1 << 42)
        comm = Recorder(rank)
        multigpu.transform_and_gather(comm, plan, in_base, stream_ptr, root_base if rank == 0 else None, size, "qtable", 0.0, 1,
                                      "compute", "comm", [Ev(k) for k in range(plan.rounds)], root=0)
        # every own plane transformed exactly once, from and into its own slot
        done = []
        for in_ptr, h, w, out_ptr, stream in launches:
            assert stream == "compute" and w == size and h % size == 0
            first = (in_ptr - in_base) // plane_in
            assert out_ptr - stream_ptr == first * plane_out
            done += list(range(lo + first, lo + first + h // size))
        assert done == list(range(lo, hi))
        # one record on the compute stream and one wait on the comm stream per round, same event, in order
        assert [e for e in fake.log if e[0] == "record"] == [("record", 1000 + k, "compute") for k in range(plan.rounds)]
        assert [e for e in fake.log if e[0] == "wait"] == [("wait", 1000 + k, "comm") for k in range(plan.rounds)]
        assert len(comm.calls) == plan.rounds and all(c[5] == 0 and c[6] == "comm" for c in comm.calls)
        per_rank[rank] = (comm.calls, stream_ptr, lo)
    landed = {p: 0 for p in range(n_planes)}
    for p in range(*plan.spans[0]):
        landed[p] += 1                                   # the root's own planes are written in place
    for k in range(plan.rounds):
        root_call = per_rank[0][0][k]
        assert root_call[1] == 0 and root_call[3][0] == 0            # root: nothing sent, nothing received from itself
        for rank in range(1, world):
            send_ptr, send_bytes, _, rb, _, _, _ = per_rank[rank][0][k]
            assert sum(rb) == 0                                        # only the root posts receives
            assert send_bytes == root_call[3][rank]                    # matched sizes: no rank waits for the other
            if send_bytes:
                first = per_rank[rank][2] + (send_ptr - per_rank[rank][1]) // plane_out
                assert root_call[4][rank] == first * plane_out         # lands at that plane's place in the batch's stream
                for p in range(first, first + send_bytes // plane_out):
                    landed[p] += 1
    assert all(v == 1 for v in landed.values())


CTL_WORKER = r'''
import os, sys
sys.path.insert(0, %(pkg)r)
from jpegx.multigpu import ControlPlane, rank_env
rank, local_rank, world = rank_env()
with ControlPlane() as ctl:
    assert ctl.allgather({"r": rank, "pid": os.getpid()})[rank]["r"] == rank
    assert [v["r"] for v in ctl.allgather({"r": rank})] == list(range(world))
    assert ctl.allreduce_max(float(rank)) == float(world - 1)
    assert ctl.allreduce_min(float(rank)) == 0.0
    ident = ctl.bcast_bytes(bytes(range(128)) if rank == 0 else None)
    assert ident == bytes(range(128))
    assert ctl.all_ok(True) is True
    assert ctl.all_ok(rank != world - 1) is False          # one failing rank is seen by everybody
    for _ in range(50):
        ctl.barrier()
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(3)
'''


def _write(tmp_path, name, text):
    path = tmp_path / name
    path.write_text(text % {"repo": REPO, "pkg": PKG})
    return str(path)


def test_own_launcher_and_control_plane(tmp_path):
    from jpegx.multigpu import launch_ranks
    script = _write(tmp_path, "ctl_worker.py", CTL_WORKER)
    env = {"JPEGX_CTL_DIR": str(tmp_path)}
    assert launch_ranks(3, [script], extra_env=env) == 0
    assert launch_ranks(2, [script, "--fail-rank", "1"], extra_env=env) == 3      # a failing rank fails the job
    assert not [f for f in os.listdir(tmp_path) if f.startswith("jpegx_ctl_")]     # rendezvous files are removed


DYING_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(pkg)r)
from jpegx.multigpu import ControlPlane, ControlPlaneError, rank_env
rank, local_rank, world = rank_env()
ctl = ControlPlane(timeout=20.0)
ctl.barrier()
if rank == 1:
    os._exit(7)                       # dies without saying goodbye, between two collectives
t0 = time.time()
try:
    ctl.barrier()
    ctl.barrier()
except ControlPlaneError:
    sys.exit(0 if time.time() - t0 < 15 else 5)      # the survivors notice promptly instead of hanging
sys.exit(4)
'''


def test_a_dead_rank_is_an_error_on_the_others_not_a_hang(tmp_path):
    import time
    from jpegx.multigpu import launch_ranks
    script = _write(tmp_path, "dying_worker.py", DYING_WORKER)
    t0 = time.time()
    rc = launch_ranks(3, [script], extra_env={"JPEGX_CTL_DIR": str(tmp_path)}, grace_s=30.0)
    assert rc == 7                     # the job fails with the dead rank's code; ranks 0 and 2 exited 0 on their own
    assert time.time() - t0 < 60


def test_control_plane_ignores_a_stale_rendezvous_file(tmp_path):
    """A file left behind by an earlier job on the same MASTER_PORT points at a dead port: the ranks
    keep polling until the live rank 0 has replaced it."""
    from jpegx.multigpu import launch_ranks
    script = _write(tmp_path, "ctl_worker.py", CTL_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        dead = s.getsockname()[1]
    port = 29876
    stale = tmp_path / ("jpegx_ctl_%d_%d_none" % (os.getuid(), port))
    stale.write_text("%d 1\n" % dead)
    # launch_ranks picks its own MASTER_PORT; force the key so that the stale file is the one consulted
    code = "import sys; sys.path.insert(0, %r)\nfrom jpegx.multigpu import launch_ranks\n" \
           "import jpegx.multigpu as m\nm._free_port = lambda: %d\n" \
           "sys.exit(launch_ranks(2, [%r], extra_env={'JPEGX_CTL_DIR': %r}))\n" % (PKG, port, script, str(tmp_path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rank_environment_is_set_before_anything_loads_hip(tmp_path):
    """What a rank needs in its environment whoever launched it: dmabuf IPC for RCCL, rocm_smi's mutex process-local
    (the cause of round 2's hang in communicator creation), phase logging; an explicit choice of the caller wins."""
    from jpegx.multigpu import rank_process_env
    env = rank_process_env({})
    assert env == {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "RSMI_MUTEX_THREAD_ONLY": "1", "JPEGX_COMM_LOG": "1"}
    assert rank_process_env({"RSMI_MUTEX_THREAD_ONLY": "0"})["RSMI_MUTEX_THREAD_ONLY"] == "0"
    # bench.py as a rank of an external launcher sets it itself, before importing jpegx
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys, runpy\nsys.argv = [%r, '--gpus', '1', '--dry-run']\n"
                     "os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
                     "for k in ('HSA_ENABLE_IPC_MODE_LEGACY', 'RSMI_MUTEX_THREAD_ONLY'): os.environ.pop(k, None)\n"
                     "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\n"
                     "print('ENV', os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), os.environ.get('RSMI_MUTEX_THREAD_ONLY'))\n"
                     % (os.path.join(REPO, "bench.py"), os.path.join(REPO, "bench.py")))
    res = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, timeout=300)
    assert "ENV 0 1" in res.stdout + res.stderr, res.stdout[-1000:] + res.stderr[-1000:]   # bench.py points fd 1 at stderr


def test_rsmi_shm_report_flags_a_mutex_left_locked_by_a_dead_thread(tmp_path):
    """The detector behind the warning printed before communicator creation: a pthread_mutex_t image whose lock word
    is set and whose owner thread does not exist is stale; an unlocked one and one held by a live thread are not."""
    import struct
    from jpegx.multigpu import rsmi_shm_report
    dead = 0x3FFFFF00                                        # no such thread
    (tmp_path / "rocm_smi_renderD1").write_bytes(struct.pack("<IIi", 1, 1, dead) + bytes(28))
    (tmp_path / "rocm_smi_renderD2").write_bytes(bytes(40))
    (tmp_path / "rocm_smi_renderD3").write_bytes(struct.pack("<IIi", 1, 1, os.getpid()) + bytes(28))
    (tmp_path / "unrelated").write_bytes(b"x")
    rep = {r["file"]: r for r in rsmi_shm_report(str(tmp_path))}
    assert sorted(rep) == ["rocm_smi_renderD1", "rocm_smi_renderD2", "rocm_smi_renderD3"]
    assert rep["rocm_smi_renderD1"]["stale"] and not rep["rocm_smi_renderD2"]["stale"] and not rep["rocm_smi_renderD3"]["stale"]
    assert rsmi_shm_report(str(tmp_path / "missing")) == []


INTRUDER_WORKER = r'''
import os, sys, socket, json, struct, time
sys.path.insert(0, %(pkg)r)
from jpegx.multigpu import ControlPlane, rank_env
rank, local_rank, world = rank_env()
if rank == 1:
    # before joining properly, play a foreign process that knows port, magic, key and world but not the token
    path = os.path.join(os.environ["JPEGX_CTL_DIR"], "jpegx_ctl_%%d_%%s_none" %% (os.getuid(), os.environ["MASTER_PORT"]))
    for _ in range(400):
        if os.path.exists(path):
            break
        time.sleep(0.05)
    port = int(open(path).read().split()[0])
    mode = os.stat(path).st_mode & 0o777
    s = socket.create_connection(("127.0.0.1", port), timeout=5)
    msg = json.dumps({"magic": ControlPlane.MAGIC, "key": "%%s_none" %% os.environ["MASTER_PORT"], "world": world, "rank": 1, "token": "guess"}).encode()
    s.sendall(struct.pack(">I", len(msg)) + msg)
    n = struct.unpack(">I", s.recv(4))[0]
    refused = json.loads(s.recv(n).decode()) == {"ok": False}
    s.close()
    if not refused or mode != 0o600:
        os._exit(9)
with ControlPlane() as ctl:
    assert ctl.allgather(rank) == list(range(world))
'''


def test_control_plane_refuses_a_hello_without_the_job_token(tmp_path):
    """The rendezvous file is private (0600, created with O_EXCL | O_NOFOLLOW) and carries a random token that a
    joining process must echo: knowing port, magic, key and world is not enough to impersonate a rank."""
    from jpegx.multigpu import launch_ranks
    script = _write(tmp_path, "intruder_worker.py", INTRUDER_WORKER)
    assert launch_ranks(2, [script], extra_env={"JPEGX_CTL_DIR": str(tmp_path)}) == 0


def _bench_line(res, returncode=0):
    assert res.returncode == returncode, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_launches_itself_for_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with nothing around it: it must start its own two ranks."""
    env = dict(os.environ, JPEGX_CTL_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--planes-total", "5"],
                         env=env, capture_output=True, text=True, timeout=300)
    line = _bench_line(res)
    assert line["dry_run"] and line["n_gpus"] == 2 and line["all_ok"] is True
    assert line["shards"] == [[0, 3], [3, 5]] and line["launcher"] == "jpegx.multigpu.launch_ranks"


def test_bench_dry_run_with_eight_ranks(tmp_path):
    """The driver's largest case as control flow only: 8 ranks, the 1024-plane batch, 128 planes each."""
    env = dict(os.environ, JPEGX_CTL_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300)
    line = _bench_line(res)
    assert line["n_gpus"] == 8 and line["all_ok"] is True and line["max_rank"] == 7.0
    assert line["shards"] == [[128 * r, 128 * (r + 1)] for r in range(8)]


def test_bench_runs_as_ranks_of_torch_distributed_run(tmp_path):
    """The driver's command for N > 1: torch.distributed.run starts the ranks, bench.py joins as one."""
    port = _free_port()
    env = dict(os.environ, JPEGX_CTL_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--planes-total", "1024"]
    line = _bench_line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600))
    assert line["dry_run"] and line["n_gpus"] == 2 and line["shards"] == [[0, 512], [512, 1024]]
    assert line["launcher"] == "external"


@pytest.mark.parametrize("launcher", ["own", "torch"])
def test_watchdog_of_the_gather_legs_keeps_the_result_line(tmp_path, launcher):
    """An exchange that never comes back (rehearsed with --dry-run-stall): rank 0 still prints its line, with the
    timeout recorded under "gather", and then every rank leaves with a NON-ZERO status, so that the job as a whole
    reports the failed exchange -- under the own launcher (which passes the ranks' status on) and as ranks of
    torch.distributed.run (which answers a failed rank with status 1)."""
    env = dict(os.environ, JPEGX_CTL_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    tail = [os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--dry-run-stall", "120", "--gather-timeout", "1"]
    if launcher == "own":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    t0 = time.time()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = _bench_line(res, returncode=3 if launcher == "own" else 1)
    assert line["dry_run"] and line["n_gpus"] == 2 and "did not finish within 1 s" in line["gather"]["error"]
    assert time.time() - t0 < 100          # nobody sat out the 120 s stall


GLOO_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(pkg)r)
import oracle
from jpegx import synth
from jpegx.multigpu import GatherPlan, shard_block_rows, shard_planes

def gather_to_root(local):
    """rank-ordered gather of unequal int16 spans (the layout jpegx_comm_gather_bytes produces)"""
    flat = torch.from_numpy(np.ascontiguousarray(local).reshape(-1).view(np.uint8))
    sizes = torch.zeros(world, dtype=torch.int64); sizes[rank] = flat.numel()
    dist.all_reduce(sizes)
    cap = int(sizes.max())
    pad = torch.cat([flat, flat.new_zeros(cap - flat.numel())])
    bufs = [torch.empty(cap, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return np.concatenate([b[:int(n)].numpy() for b, n in zip(bufs, sizes)]).view(np.int16)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ok = True
# (1) a batch of 5 independent planes (uneven split), laid out by GatherPlan ---------------------
lo, hi = shard_planes(5, world, rank)
local = [oracle.forward_f32(synth.generate_plane("noise", 32, 64, seed=0, plane=p), "qtable") for p in range(lo, hi)]
got = gather_to_root(np.stack(local))
if rank == 0:
    full = np.stack([oracle.forward_f32(synth.generate_plane("noise", 32, 64, seed=0, plane=p), "qtable") for p in range(5)])
    ok = ok and np.array_equal(got.reshape(full.shape), full)
    plan = GatherPlan(5, world, full[0].nbytes, 2)
    ok = ok and plan.spans == [shard_planes(5, world, r) for r in range(world)]
# (2) block-row ranges of one tall plane --------------------------------------------------------
H, W = 88, 64
y0, y1 = shard_block_rows(H, world, rank)
plane = synth.generate_plane("smooth", H, W, seed=3)
got = gather_to_root(oracle.forward_f32(plane[y0:y1], "qtable"))
if rank == 0:
    ok = ok and np.array_equal(got.reshape(H // 8, W // 8, 64), oracle.forward_f32(plane, "qtable"))
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = _write(tmp_path, "gloo_worker.py", GLOO_WORKER)
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
