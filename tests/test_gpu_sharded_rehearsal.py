"""The rank grid of bench.py with the REAL engine: 4 ranks share this box's GPU and exchange over
gloo (EXPANN_BENCH_REHEARSAL), rank 0 compares the sharded result with an unsharded search bit for
bit (bench.py --verify).  The RCCL transport itself only exists on a multi-GPU node; everything
above it -- shard ranges, id offsets, chunk layout, strided merge, the two collectives -- runs here."""
import json
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
@pytest.mark.parametrize("ranks,row_shards,k,exchange,pattern",
                         [(4, 0, 10, "auto", "auto"), (2, 0, 10, "auto", "auto"), (4, 2, 10, "auto", "auto"),
                          (4, 4, 100, "auto", "auto"), (4, 2, 100, "auto", "auto"), (3, 3, 100, "auto", "slices"),
                          (4, 0, 10, "native", "auto"), (3, 0, 100, "native", "slices"),
                          (3, 0, 10, "native", "allgather")])
def test_rank_grid_with_the_real_engine(ranks, row_shards, k, exchange, pattern):
    """row_shards 0 = bench.py's default (pure row sharding, every rank searches every query);
    k = 100 is BASELINE configs[2]'s.  The deferred check stays on (searches, merge and exchange are
    enqueued back to back on the work stream, one expann_sync per timed region).
    exchange "native" = the rank form of the C ABI itself (expann_sharded_create_rank without an RCCL
    id, expann_sharded_search_device: local scan -> exchange -> merge inside the library) with the
    transport handed in through expann_sharded_set_alltoallv_fn / _set_exchange_fn, since RCCL refuses
    ranks that share a GPU; "auto" = torch.distributed above the single-device handle.  pattern: slices
    (the sharded handle's default) = all-to-all of query slices + all-gather of the merged slices;
    allgather = one all-gather of whole chunks."""
    env = dict(os.environ, EXPANN_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--row-shards", str(row_shards),
           "--rows", "200000", "--queries", "1000", "--k", str(k), "--steps", "1", "--warmup", "0", "--verify",
           "--no-cpu-baseline", "--exchange", exchange, "--exchange-pattern", pattern]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "IDENTICAL to the unsharded search" in res.stderr, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == ranks and line["bit_exact"] and "rehearsal" in line["config"]
    if exchange == "native":
        assert "the caller's transport (gloo)" in line["config"]["sharding"], line["config"]["sharding"]
        want = "one all-gather of whole" if pattern == "allgather" else "all-to-all of query slices"
        assert want in line["config"]["sharding"], line["config"]["sharding"]


def _plain_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EXPANN_BENCH_REHEARSAL")}
    env.update(extra)
    return env


@pytest.mark.parametrize("gpus,k,pattern,rows", [(4, 10, "auto", 200000), (3, 100, "allgather", 200000), (8, 10, "slices", 200000),
                                                 (4, 3, "auto", 5)])
def test_plain_launch_drives_all_shards_in_one_process(gpus, k, pattern, rows):
    """`python bench.py --gpus G` WITHOUT a launcher: one process, the in-process handle over G devices
    (expann_sharded_create + _set_shard_device + _search_devices; here the G shards share this box's GPU under
    EXPANN_BENCH_REHEARSAL and exchange by device copies).  The JSON line must say what ran: n_gpus = G, the
    launch form, the rows per rank, the communicator's size (0: no RCCL communicator on one device).
    rows = 5 over 4 shards: 2, 2, 1, 0 -- the shard without rows is left out."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--rows", str(rows), "--queries", "1000",
           "--k", str(k), "--steps", "2", "--warmup", "1", "--verify", "--no-cpu-baseline", "--exchange-pattern", pattern]
    res = subprocess.run(cmd, env=_plain_env(EXPANN_BENCH_REHEARSAL="1"), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "IDENTICAL to the unsharded search" in res.stderr, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert line["n_gpus"] == gpus and line["bit_exact"] and line["recall_measured"] == 1.0
    assert "one process, one handle over all devices" in cfg["launch"] and "rehearsal" in cfg
    assert sum(cfg["rows_per_rank"]) == rows and len(cfg["rows_per_rank"]) == gpus and cfg["comm_ranks"] == 0
    assert "device copies" in cfg["sharding"]
    assert ("one all-gather of whole" if pattern == "allgather" else "all-to-all of query slices") in cfg["sharding"]
    assert cfg["host_enqueue_ms_per_step"] > 0


def test_plain_launch_refuses_more_gpus_than_the_box_has():
    """outside the rehearsal `--gpus G` with fewer than G visible devices exits non-zero and prints no line"""
    import torch
    have = torch.cuda.device_count()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--rows", "100000",
                          "--queries", "500", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=_plain_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert f"only {have} HIP device(s) visible" in res.stderr, res.stderr[-2000:]


def test_plain_launch_one_gpu_is_the_headline_path():
    """`python bench.py --gpus 1` (what the driver runs) on a reduced shape: the single-device handle, no
    sharding, the fields the judge reads."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rows", "300000", "--queries",
                          "2000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "2"],
                         env=_plain_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["sharding"] == "none" and line["config"]["comm_ranks"] == 0
    assert line["config"]["launch"] == "one process, one GPU" and line["bit_exact"] and line["recall_measured"] == 1.0
    assert line["roofline"]["kernel"].startswith("scan_gemm_f16x") and "cpu_baseline" in line
    assert "verified_step" in line
