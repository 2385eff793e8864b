"""Multi-GPU path on the CPU: world_size-2 gloo run of the sharding + gather logic
(jpegx/multigpu.py).  The per-rank transform is stood in for by the oracle (no GPU here); what is
under test is that contiguous plane / block-row shards plus the rank-ordered gather reproduce the
un-sharded coefficient stream exactly."""
import os
import socket
import subprocess
import sys

import numpy as np
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


WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(pkg)r)
import oracle
from jpegx import synth
from jpegx.multigpu import gather_stream, shard_block_rows, shard_planes

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# (1) a batch of 5 independent planes (uneven split) -------------------------------------------
lo, hi = shard_planes(5, world, rank)
local = [oracle.forward_f32(synth.generate_plane("noise", 32, 64, seed=0, plane=p), "qtable") for p in range(lo, hi)]
local = torch.from_numpy(np.stack(local)) if local else torch.empty(0, dtype=torch.int16)
parts = gather_stream(local, dst=0)
ok = True
if rank == 0:
    full = np.stack([oracle.forward_f32(synth.generate_plane("noise", 32, 64, seed=0, plane=p), "qtable") for p in range(5)])
    got = torch.cat(parts).numpy().reshape(full.shape)
    ok = ok and np.array_equal(got, full)
else:
    ok = ok and parts is None
# (2) block-row ranges of one tall plane --------------------------------------------------------
H, W = 88, 64
y0, y1 = shard_block_rows(H, world, rank)
plane = synth.generate_plane("smooth", H, W, seed=3)
mine = torch.from_numpy(oracle.forward_f32(plane[y0:y1], "qtable"))
parts = gather_stream(mine, dst=0)
if rank == 0:
    got = torch.cat(parts).numpy().reshape(H // 8, W // 8, 64)
    ok = ok and np.array_equal(got, oracle.forward_f32(plane, "qtable"))
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO, "pkg": PKG})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
