"""world_size-2 (and 3) test of the multi-GPU orchestration over gloo on CPU: shard ranges, id
offsets, all-gather layout and merge order.  The local search and the merge are CPU stand-ins
built on the oracle (the production ones are HIP kernels, covered by
tests/test_gpu_full_size.py::test_c2_shard_and_merge_equals_unsharded)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _merge_np(all_ids, all_d):
    """numpy statement of merge_topk_kernel: per query the k smallest (score, id) of G lists."""
    G, m, k = all_ids.shape
    out_i = np.empty((m, k), np.int64)
    out_d = np.empty((m, k), np.float32)
    for q in range(m):
        ids = all_ids[:, q, :].reshape(-1).astype(np.uint64)
        d = all_d[:, q, :].reshape(-1)
        order = np.lexsort((ids, d))[:k]
        out_i[q] = ids[order].astype(np.int64)
        out_d[q] = d[order]
    return out_i, out_d


def _worker(rank, world, port, n, d, m, k, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle_ctypes as oc
    from expann_amd.sharded import ShardedSearch, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(77)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[n // 3] = base[2 * n // 3]          # a cross-shard exact tie
    queries = rng.standard_normal((m, d)).astype(np.float32)
    queries[0] = base[n // 3]
    lo, hi = shard_range(n, rank, world)

    def local_search(q, kk):
        ids, dd = oc.brute_force(base[lo:hi], q.numpy(), kk)
        ids = np.where(ids == np.uint64(2 ** 64 - 1), ids, ids + np.uint64(lo))
        return torch.from_numpy(ids.view(np.int64)), torch.from_numpy(dd)

    def merge(all_ids, all_d):
        i, dd = _merge_np(all_ids.numpy(), all_d.numpy())
        return torch.from_numpy(i), torch.from_numpy(dd)

    ss = ShardedSearch(dist, world, local_search, merge,
                       lambda t: torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype))
    ids, dd = ss.search(torch.from_numpy(queries), k)
    ref_ids, ref_d = oc.brute_force(base, queries, k)
    ok = np.array_equal(ids.numpy().view(np.uint64), ref_ids) and np.array_equal(dd.numpy(), ref_d)
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _grid_worker(rank, world, row_shards, port, n, d, m, k, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle_ctypes as oc
    from expann_amd.sharded import GridShardedSearch, shard_range, shard_grid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(78)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[n // 3] = base[2 * n // 3]          # a cross-shard exact tie
    queries = rng.standard_normal((m, d)).astype(np.float32)
    queries[0] = base[n // 3]
    R, Q = shard_grid(world, row_shards)
    lo, hi = shard_range(n, rank % R, R)
    pad = (m + Q - 1) // Q
    bufs = {}
    from expann_amd.sharded import unpack_chunk, chunk_bytes

    def alloc(name, nbytes, like):
        if name not in bufs:
            bufs[name] = torch.zeros(nbytes, dtype=torch.uint8)
        return bufs[name]

    def local_search(q, kk, chunk):
        ids_v, d_v = unpack_chunk(chunk, pad, kk)
        ids_v.fill_(-1)
        d_v.fill_(float("inf"))
        i, x = oc.brute_force(base[lo:hi], q.numpy(), kk)
        i = np.where(i == np.uint64(2 ** 64 - 1), i, i + np.uint64(lo))
        ids_v[:len(i)] = torch.from_numpy(i.view(np.int64))
        d_v[:len(i)] = torch.from_numpy(x)

    def merge(gathered, n_lists, rows, kk, out_chunk):
        cb = chunk_bytes(rows, kk)
        lists = [unpack_chunk(gathered[g * cb:(g + 1) * cb], rows, kk) for g in range(n_lists)]
        all_ids = np.stack([l[0].numpy() for l in lists], 0)
        all_d = np.stack([l[1].numpy() for l in lists], 0)
        i, dd = _merge_np(all_ids, all_d)
        oi, od = unpack_chunk(out_chunk, rows, kk)
        oi.copy_(torch.from_numpy(i))
        od.copy_(torch.from_numpy(dd))

    gs = GridShardedSearch(dist, world, rank, row_shards, local_search, merge, alloc)
    ids, dd = gs.search(torch.from_numpy(queries), k)
    ref_ids, ref_d = oc.brute_force(base, queries, k)
    ok = ids.shape[0] == m and np.array_equal(ids.numpy().view(np.uint64), ref_ids) and \
        np.array_equal(dd.numpy(), ref_d)
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,row_shards,m", [(4, 2, 10), (4, None, 9), (2, None, 7), (6, 2, 10),
                                                (3, None, 5), (2, 1, 7), (8, None, 12), (8, 2, 11)])
def test_grid_sharded_search_matches_unsharded(world, row_shards, m):
    """rows x queries grids (None = bench.py's default, pure row sharding; 8 ranks = the C3 node):
    every rank must end with the full,
    unsharded answer, ragged query slices included."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, world, row_shards, port, 2003, 64, m, 10, out))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(out.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def test_shard_grid_defaults():
    from expann_amd.sharded import shard_grid
    assert shard_grid(1) == (1, 1)
    assert shard_grid(2) == (2, 1)
    assert shard_grid(4) == (4, 1)      # default: pure row sharding
    assert shard_grid(8) == (8, 1)
    assert shard_grid(8, 2) == (2, 4)   # hybrid grid on request
    assert shard_grid(3) == (3, 1)
    assert shard_grid(8, 8) == (8, 1)
    with pytest.raises(ValueError):
        shard_grid(8, 3)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_matches_unsharded(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 3001, 64, 9, 10, out))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(out.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def test_shard_ranges_partition_the_base():
    from expann_amd.sharded import shard_range
    for n in (1, 7, 1000, 1_000_000, 10_000_019):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def _slice_worker(rank, world, port, n, d, m, k, out):
    """exchange pattern 2 (all-to-all of query slices + all-gather of the merged slices) over gloo, CPU
    stand-ins for the scan and the merge; the ceil partition of the ROWS as the C ABI cuts them (trailing
    ranks may hold no rows and then contribute all-padding lists)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import oracle_ctypes as oc
    from expann_amd.sharded import SliceShardedSearch, ceil_shard_range, slice_range, unpack_chunk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(79)
    base = rng.standard_normal((n, d)).astype(np.float32)
    if n > 3:
        base[n // 3] = base[2 * n // 3]          # a cross-shard exact tie
    queries = rng.standard_normal((m, d)).astype(np.float32)
    queries[0] = base[n // 3]
    lo, hi = ceil_shard_range(n, rank, world)
    bufs = {}

    def alloc(name, nbytes, like):
        if name not in bufs:
            bufs[name] = torch.zeros(nbytes, dtype=torch.uint8)
        return bufs[name]

    def local_search(q, kk, chunk):
        ids_v, d_v = unpack_chunk(chunk, q.shape[0], kk)
        ids_v.fill_(-1)
        d_v.fill_(float("inf"))
        if hi > lo:
            i, x = oc.brute_force(base[lo:hi], q.numpy(), kk)
            i = np.where(i == np.uint64(2 ** 64 - 1), i, i + np.uint64(lo))
            ids_v.copy_(torch.from_numpy(i.view(np.int64)))
            d_v.copy_(torch.from_numpy(x))

    def merge_slices(l_ids, l_d, n_lists, per, cnt, kk, o_ids, o_d):
        all_ids = np.stack([l_ids[g * per:g * per + cnt].numpy() for g in range(n_lists)], 0)
        all_d = np.stack([l_d[g * per:g * per + cnt].numpy() for g in range(n_lists)], 0)
        i, dd = _merge_np(all_ids, all_d)
        o_ids.copy_(torch.from_numpy(i))
        o_d.copy_(torch.from_numpy(dd))

    ss = SliceShardedSearch(dist, world, rank, local_search, merge_slices, alloc)
    ids, dd = ss.search(torch.from_numpy(queries), k)
    ref_ids, ref_d = oc.brute_force(base, queries, k)
    ok = ids.shape[0] == m and np.array_equal(ids.numpy().view(np.uint64), ref_ids) and \
        np.array_equal(dd.numpy(), ref_d)
    # the slices partition the queries
    edges = [slice_range(m, j, world) for j in range(world)]
    ok = ok and edges[0][0] == 0 and edges[-1][1] == m and all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,m,k", [(2, 2003, 7, 10), (3, 2003, 10, 10), (4, 2003, 9, 100), (8, 2003, 12, 10),
                                         (4, 5, 6, 3), (8, 2003, 5, 10)])
def test_slice_exchange_matches_unsharded(world, n, m, k):
    """pattern 2 of csrc/expann_sharded.hip, restated over gloo: ragged query slices, more ranks than
    queries (empty slices), a rank without rows (n = 5 over 4 ranks: 2, 2, 1, 0), k = 100 > rows of a shard."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slice_worker, args=(r, world, port, n, 64, m, k, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(out.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


def test_bench_refuses_to_run_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` without a launcher drives 2 devices in one process (the in-process handle)
    or fails loudly -- on this GPU-less box it must fail, never fall through to one GPU; so must a launcher
    whose WORLD_SIZE disagrees with --gpus."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EXPANN_BENCH_REHEARSAL")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    if res.returncode == 0:
        pytest.skip("a GPU box with >= 2 devices ran it")
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "--gpus 2" in res.stderr and ("HIP device" in res.stderr)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in res.stderr and res.stdout.strip() == ""
