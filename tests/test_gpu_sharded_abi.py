"""The row-sharded engine behind the C ABI (expann_sharded_*, csrc/expann_sharded.hip) on the
one-GPU test box: several shards on device 0 (exchange = device copies; RCCL refuses duplicate
devices), one-rank RCCL communicators (ncclCommInitAll / ncclCommInitRank + ncclAllGather really
run), BASELINE configs[2]'s shape per GPU (1.25 M rows x d128, k = 100).  In every case ids and
distance bits must equal the single-device index and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _plain(base, q, k, metric="l2", dtype="f32"):
    from expann_amd import GpuBruteForceEngine
    e = GpuBruteForceEngine(base.shape[1], metric, dtype)
    e.store_many_vectors(base)
    e.build()
    out = e.query_k_batch(q, k)
    e.close()
    return out


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.mark.parametrize("n_dev,exchange,want", [(1, 0, 0), (1, 1, 1), (8, 0, 2), (3, 2, 2)])
def test_inprocess_shards_equal_the_plain_index(oracle, n_dev, exchange, want):
    from expann_amd import ShardedBruteForceEngine
    rng = np.random.RandomState(5 + n_dev)
    n, d, m, k = 150_001, 128, 700, 10
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[70_000:70_040] = base[3]          # exact ties across shard boundaries
    q = rng.standard_normal((m, d)).astype(np.float32)
    q[5] = base[3]
    eng = ShardedBruteForceEngine(d, "l2", "f32", devices=[0] * n_dev)
    if exchange:
        eng.set_option("exchange", exchange)
    eng.store_many_vectors(base[:40_000])
    eng.store_many_vectors(base[40_000:])
    eng.build()
    assert eng.size() == n and eng.shards() == n_dev and eng.exchange() == want
    got = eng.query_k_batch(q, k)
    assert _same(got, _plain(base, q, k))
    ref = oracle.brute_force(base, q[:64], k, oracle.METRIC_L2_F32, n_threads=8)
    assert _same((got[0][:64], got[1][:64]), ref)
    assert eng.query_k(q[5], 4) == [int(x) for x in ref[0][5][:4]]
    # a second batch of another shape reuses the handle (buffers regrow), k = 100
    got100 = eng.query_k_batch(q[:90], 100)
    assert _same(got100, _plain(base, q[:90], 100))
    eng.close()


def test_fewer_rows_than_shards_and_padding(oracle):
    """N = 50 rows over 8 shards (ceil partition: 7 rows each, the last holds 1), k = 10 > rows per
    shard (per-shard lists are padded), k = 64 > N (the merged list is padded)."""
    from expann_amd import ShardedBruteForceEngine
    rng = np.random.RandomState(11)
    base = rng.standard_normal((50, 64)).astype(np.float32)
    q = rng.standard_normal((9, 64)).astype(np.float32)
    eng = ShardedBruteForceEngine(64, "l2", "f32", devices=[0] * 8)
    eng.store_many_vectors(base)
    eng.build()
    assert eng.shards() == 8
    for k in (10, 64):
        got = eng.query_k_batch(q, k)
        ref = oracle.brute_force(base, q, k, oracle.METRIC_L2_F32)
        assert _same(got, ref), k
    eng.close()
    eng = ShardedBruteForceEngine(64, "l2", "f32", devices=[0] * 8)
    eng.store_many_vectors(base[:5])     # 5 rows: only 5 shards come into use
    eng.build()
    assert eng.shards() == 5
    assert _same(eng.query_k_batch(q, 3), oracle.brute_force(base[:5], q, 3, oracle.METRIC_L2_F32))
    eng.close()


def test_int8_ip_shards(oracle):
    from expann_amd import ShardedBruteForceEngine
    rng = np.random.RandomState(12)
    base = rng.randint(-127, 128, size=(70_000, 128)).astype(np.int8)
    q = rng.randint(-127, 128, size=(130, 128)).astype(np.int8)
    eng = ShardedBruteForceEngine(128, "ip", "i8", devices=[0, 0, 0, 0])
    eng.store_many_vectors(base)
    eng.build()
    got = eng.query_k_batch(q, 10)
    assert _same(got, oracle.brute_force(base, q, 10, oracle.METRIC_IP_I8, n_threads=8))
    eng.close()


def test_rank_form_with_a_one_rank_rccl_communicator(oracle):
    """expann_sharded_create_rank(rank 0 of 1) with a unique id: ncclCommInitRank, then every search
    is local scan -> ncclAllGather (one rank) -> merge on the caller's stream, deferred check on."""
    torch = pytest.importorskip("torch")
    from expann_amd import ShardedBruteForceEngine
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    n, d, m, k = 300_000, 128, 1000, 10
    base = torch.randn(n, d, device="cuda", generator=g)
    q = torch.randn(m, d, device="cuda", generator=g)
    eng = ShardedBruteForceEngine(d, "l2", "f32", device=0, rank=0, world=1,
                                  unique_id=ShardedBruteForceEngine.unique_id())
    assert eng.exchange() == 1 and eng.shards() == 1
    eng.set_shard_device(0, base.data_ptr(), n, 1_000_000)      # global ids start at 1e6
    eng.set_option("async_search", 1)
    ids = torch.empty(m, k, dtype=torch.int64, device="cuda")
    dd = torch.empty(m, k, dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        eng.search_device(q.data_ptr(), m, k, ids.data_ptr(), dd.data_ptr(), st.cuda_stream)
    eng.sync()
    torch.cuda.synchronize()
    sel = np.arange(0, m, 25)
    rids, rd = oracle.brute_force(base.cpu().numpy(), q.cpu().numpy()[sel], k, oracle.METRIC_L2_F32, n_threads=16)
    assert np.array_equal(ids.cpu().numpy().view(np.uint64)[sel], rids + np.uint64(1_000_000))
    assert np.array_equal(dd.cpu().numpy()[sel].view(np.uint32), rd.view(np.uint32))
    eng.close()


def test_c3_shape_per_gpu_k100(oracle):
    """BASELINE configs[2] as one GPU sees it: a 1.25 M-row shard of d128 fp32, 10 k queries,
    k = 100 -- searched (a) as ONE index and (b) cut into 8 shards + exchange + merge behind the C
    ABI; both bit-equal, and a sample of queries bit-equal to the oracle's scan of all rows."""
    torch = pytest.importorskip("torch")
    from expann_amd import GpuBruteForceEngine, ShardedBruteForceEngine
    n, d, m, k = 1_250_000, 128, 10_000, 100
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    base = torch.randn(n, d, device="cuda", generator=g)
    g.manual_seed(4321)
    q = torch.randn(m, d, device="cuda", generator=g)
    one = GpuBruteForceEngine(d, "l2")
    one.set_base_device(base.data_ptr(), n, 0)
    ids = torch.empty(m, k, dtype=torch.int64, device="cuda")
    dd = torch.empty(m, k, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    one.search_device(q.data_ptr(), m, k, ids.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((dd[:, 1:] >= dd[:, :-1]).all())
    srt = ids.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "duplicate ids"
    base_h, q_h = base.cpu().numpy(), q.cpu().numpy()
    sel = np.arange(0, m, m // 24)[:24]
    rids, rd = oracle.brute_force(base_h, q_h[sel], k, oracle.METRIC_L2_F32, n_threads=16)
    assert np.array_equal(ids.cpu().numpy().view(np.uint64)[sel], rids)
    assert np.array_equal(dd.cpu().numpy()[sel].view(np.uint32), rd.view(np.uint32))
    one.close()
    sh = ShardedBruteForceEngine(d, "l2", "f32", devices=[0] * 8)
    per = (n + 7) // 8
    for r in range(8):
        lo, hi = r * per, min(n, (r + 1) * per)
        sh.set_shard_device(r, base[lo:hi].data_ptr(), hi - lo, lo)
    assert sh.shards() == 8 and sh.size() == n
    sids, sd = sh.query_k_batch(q_h[:2000], k)
    assert np.array_equal(sids, ids.cpu().numpy().view(np.uint64)[:2000])
    assert np.array_equal(sd.view(np.uint32), dd.cpu().numpy()[:2000].view(np.uint32))
    sh.close()
