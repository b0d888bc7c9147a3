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


@pytest.mark.parametrize("ranks,row_shards", [(4, 0), (2, 0), (4, 4)])
def test_rank_grid_with_the_real_engine(ranks, row_shards):
    env = dict(os.environ, EXPANN_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--row-shards", str(row_shards),
           "--rows", "200000", "--queries", "1000", "--steps", "1", "--warmup", "0", "--verify", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "IDENTICAL to the unsharded search" in res.stderr, res.stderr[-2000:]
