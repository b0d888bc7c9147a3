"""The rank grid of bench.py with the REAL engine: 4 ranks share this box's GPU and exchange over
gloo (EXPANN_BENCH_REHEARSAL), rank 0 compares the sharded result with an unsharded search bit for
bit (bench.py --verify).  The RCCL transport itself only exists on a multi-GPU node; everything
above it -- shard ranges, id offsets, chunk layout, strided merge, the two collectives -- runs here."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# (at most 6 processes may have the box's GPU open, this test process and the launcher included: 4
# ranks here; the 8-rank grids of the C3 node run in tests/test_sharded_gloo.py with CPU stand-ins
# and, in one process, in tests/test_gpu_sharded_abi.py)
@pytest.mark.parametrize("ranks,row_shards,k,exchange", [(4, 0, 10, "auto"), (2, 0, 10, "auto"), (4, 2, 10, "auto"),
                                                          (4, 4, 100, "auto"), (4, 2, 100, "auto"),
                                                          (3, 3, 100, "auto"), (4, 0, 10, "native"),
                                                          (3, 0, 100, "native")])
def test_rank_grid_with_the_real_engine(ranks, row_shards, k, exchange):
    """row_shards 0 = bench.py's default (pure row sharding, every rank searches every query);
    k = 100 is BASELINE configs[2]'s.  The deferred check stays on (searches, merge and exchange are
    enqueued back to back on the work stream, one expann_sync per timed region).
    exchange "native" = the rank form of the C ABI itself (expann_sharded_create_rank without an RCCL
    id, expann_sharded_search_device: local scan -> exchange -> merge inside the library) with the
    all-gather handed in through expann_sharded_set_exchange_fn, since RCCL refuses ranks that share
    a GPU; "auto" = torch.distributed above the single-device handle."""
    env = dict(os.environ, EXPANN_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--row-shards", str(row_shards),
           "--rows", "200000", "--queries", "1000", "--k", str(k), "--steps", "1", "--warmup", "0", "--verify",
           "--no-cpu-baseline", "--exchange", exchange]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "IDENTICAL to the unsharded search" in res.stderr, res.stderr[-2000:]
    if exchange == "native":
        assert "the caller's all-gather (gloo) + merge behind the C ABI" in res.stdout, res.stdout[-2000:]
