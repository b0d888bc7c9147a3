"""GPU tests at BASELINE.json's FULL sizes: C2 (1M x d128 fp32, 10k queries, k=10), C3's whole index on
one GPU (10M x d128 fp32, k=100), C5 (10M x d768 int8 inner product, k=10) and C4 (1M-row graph in the
reference's sweep configuration, built by the batched GPU builder, ef_search = 60, both compression
modes): a sampled bit-exact check against the oracle plus size-independent properties (sortedness, no
duplicate ids -- src/basic_bench.h:98-104 --, ids in range, idempotence, shard-and-merge == unsharded)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, M, K = 1_000_000, 128, 10_000, 10


@pytest.fixture(scope="module")
def c2():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    from expann_amd import GpuBruteForceEngine
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    base = torch.randn(N, D, device="cuda", dtype=torch.float32, generator=g)
    g.manual_seed(4321)
    queries = torch.randn(M, D, device="cuda", dtype=torch.float32, generator=g)
    eng = GpuBruteForceEngine(D, "l2")
    eng.set_base_device(base.data_ptr(), N, 0)
    ids = torch.empty(M, K, dtype=torch.int64, device="cuda")   # uint64 bit pattern
    dists = torch.empty(M, K, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.search_device(queries.data_ptr(), M, K, ids.data_ptr(), dists.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dict(torch=torch, base=base, queries=queries, eng=eng, ids=ids, dists=dists)


def test_c2_properties(c2):
    torch = c2["torch"]
    ids, dists = c2["ids"], c2["dists"]
    assert bool((dists[:, 1:] >= dists[:, :-1]).all()), "distances not ascending"
    assert bool((ids >= 0).all()) and bool((ids < N).all())
    srt = ids.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "duplicate ids (basic_bench.h:98-104)"
    # distances really are the distances of the returned rows (fp32 recompute, 1e-4 rel)
    sel = torch.arange(0, M, 97, device="cuda")
    rows = c2["base"][ids[sel]]                          # [s, K, D]
    d2 = ((rows - c2["queries"][sel][:, None, :]) ** 2).sum(-1)
    assert torch.allclose(d2, dists[sel], rtol=1e-4, atol=0)
    # nothing nearer was missed: any probed row nearer than the k-th result must be in the result
    probe_idx = torch.randint(0, N, (4096,), device="cuda")
    dq = torch.cdist(c2["queries"][sel], c2["base"][probe_idx]) ** 2
    kth = dists[sel][:, -1:]
    nearer = dq < kth * (1 - 1e-4)
    member = (ids[sel][:, :, None] == probe_idx[None, None, :]).any(1)
    assert bool((~nearer | member).all())


def test_c2_idempotent(c2):
    torch = c2["torch"]
    ids2 = torch.empty_like(c2["ids"])
    d2 = torch.empty_like(c2["dists"])
    c2["eng"].search_device(c2["queries"].data_ptr(), M, K, ids2.data_ptr(), d2.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(ids2, c2["ids"]) and torch.equal(d2, c2["dists"])


def test_c2_sampled_bit_exact_vs_oracle(c2, oracle):
    """40 of the 10k queries against the full 1M-row oracle scan: ids and distance bits."""
    sel = np.arange(0, M, M // 40)[:40]
    base = c2["base"].cpu().numpy()
    q = c2["queries"].cpu().numpy()[sel]
    rids, rd = oracle.brute_force(base, q, K, oracle.METRIC_L2_F32, n_threads=16)
    ids = c2["ids"].cpu().numpy().view(np.uint64)[sel]
    dists = c2["dists"].cpu().numpy()[sel]
    assert np.array_equal(ids, rids)
    assert np.array_equal(dists.view(np.uint32), rd.view(np.uint32))
    assert oracle.recall(ids, rids) == 1.0


def test_c2_shard_and_merge_equals_unsharded(c2):
    """Two half-base shards with id offsets + expann_merge_topk_device == the unsharded result
    (the multi-GPU path of bench.py, SURVEY 8e, on one device)."""
    torch = c2["torch"]
    from expann_amd import GpuBruteForceEngine, merge_topk_device
    m = 2000
    q = c2["queries"][:m].contiguous()
    halves = [(0, N // 2 + 13), (N // 2 + 13, N)]
    all_ids = torch.empty(2, m, K, dtype=torch.int64, device="cuda")
    all_d = torch.empty(2, m, K, dtype=torch.float32, device="cuda")
    engs = []
    for r, (lo, hi) in enumerate(halves):
        e = GpuBruteForceEngine(D, "l2")
        e.set_base_device(c2["base"][lo:hi].data_ptr(), hi - lo, lo)
        e.search_device(q.data_ptr(), m, K, all_ids[r].data_ptr(), all_d[r].data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
        engs.append(e)
    out_ids = torch.empty(m, K, dtype=torch.int64, device="cuda")
    out_d = torch.empty(m, K, dtype=torch.float32, device="cuda")
    merge_topk_device(0, all_ids.data_ptr(), all_d.data_ptr(), 2, m, K, out_ids.data_ptr(),
                      out_d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out_ids, c2["ids"][:m]) and torch.equal(out_d, c2["dists"][:m])
    for e in engs:
        e.close()


def _properties(torch, ids, dists, n):
    assert bool((dists[:, 1:] >= dists[:, :-1]).all()), "scores not ascending"
    assert bool((ids >= 0).all()) and bool((ids < n).all()), "id out of range"
    srt = ids.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "duplicate ids (basic_bench.h:98-104)"


def test_c3_whole_index_on_one_gpu(oracle):
    """BASELINE configs[2]'s index unsharded: 10M x d128 fp32, 10k queries, k = 100 (what `bench.py
    --workload c3` runs on one GPU; 8 GPUs hold 1.25 M rows each: tests/test_gpu_sharded_abi.py).
    Properties over all 10^6 results + 16 queries bit-equal to the oracle's scan of all 10M rows."""
    torch = pytest.importorskip("torch")
    from expann_amd import GpuBruteForceEngine
    n, d, m, k = 10_000_000, 128, 10_000, 100
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    base = torch.randn(n, d, device="cuda", dtype=torch.float32, generator=g)
    g.manual_seed(4321)
    q = torch.randn(m, d, device="cuda", dtype=torch.float32, generator=g)
    eng = GpuBruteForceEngine(d, "l2")
    eng.set_base_device(base.data_ptr(), n, 0)
    ids = torch.empty(m, k, dtype=torch.int64, device="cuda")
    dd = torch.empty(m, k, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    eng.search_device(q.data_ptr(), m, k, ids.data_ptr(), dd.data_ptr(), st)
    torch.cuda.synchronize()
    _properties(torch, ids, dd, n)
    ids2, dd2 = torch.empty_like(ids), torch.empty_like(dd)
    eng.search_device(q.data_ptr(), m, k, ids2.data_ptr(), dd2.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(ids, ids2) and torch.equal(dd, dd2), "not idempotent"
    # the k-th distance bounds every probed row that is not in the result
    sel = torch.arange(0, m, 199, device="cuda")
    probe = torch.randint(0, n, (2048,), device="cuda")
    dq = torch.cdist(q[sel], base[probe]) ** 2
    member = (ids[sel][:, :, None] == probe[None, None, :]).any(1)
    assert bool((~(dq < dd[sel][:, -1:] * (1 - 1e-4)) | member).all())
    sel_h = np.arange(0, m, m // 16)[:16]
    base_h = base.cpu().numpy()
    rids, rd = oracle.brute_force(base_h, q.cpu().numpy()[sel_h], k, oracle.METRIC_L2_F32, n_threads=16)
    assert np.array_equal(ids.cpu().numpy().view(np.uint64)[sel_h], rids)
    assert np.array_equal(dd.cpu().numpy()[sel_h].view(np.uint32), rd.view(np.uint32))
    eng.close()


def test_c5_int8_inner_product_full_size(oracle):
    """BASELINE configs[4]: 10M x d768 int8, inner product, 10k queries, k = 10 (int8 MFMA filter, exact
    integer scores): properties + 16 queries bit-equal to the oracle's scan of all 10M rows."""
    torch = pytest.importorskip("torch")
    from expann_amd import GpuBruteForceEngine
    n, d, m, k = 10_000_000, 768, 10_000, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    base = torch.randint(-127, 128, (n, d), device="cuda", dtype=torch.int8, generator=g)
    g.manual_seed(4321)
    q = torch.randint(-127, 128, (m, d), device="cuda", dtype=torch.int8, generator=g)
    eng = GpuBruteForceEngine(d, "ip", "i8")
    eng.set_base_device(base.data_ptr(), n, 0)
    eng.set_profiling(True)
    ids = torch.empty(m, k, dtype=torch.int64, device="cuda")
    dd = torch.empty(m, k, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.search_device(q.data_ptr(), m, k, ids.data_ptr(), dd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert eng.get_profile()["scan_kernel"].startswith("scan_gemm_i8"), "the int8 MFMA form did not run"
    _properties(torch, ids, dd, n)
    # the scores are the exact integers -q.b of the returned rows
    sel = torch.arange(0, m, 499, device="cuda")
    dots = (base[ids[sel]].to(torch.int32) * q[sel][:, None, :].to(torch.int32)).sum(-1)
    assert torch.equal(dd[sel], (-dots).to(torch.float32))
    sel_h = np.arange(0, m, m // 16)[:16]
    rids, rd = oracle.brute_force(base.cpu().numpy(), q.cpu().numpy()[sel_h], k, oracle.METRIC_IP_I8, n_threads=16)
    assert np.array_equal(ids.cpu().numpy().view(np.uint64)[sel_h], rids)
    assert np.array_equal(dd.cpu().numpy()[sel_h].view(np.uint32), rd.view(np.uint32))
    eng.close()


def test_c4_graph_one_million_rows(oracle, tmp_path):
    """BASELINE configs[3] at its size: 1M SIFT-like rows (SURVEY 8d's stand-in), the reference's sweep
    configuration (src/bench_runner.h:133-162: M = 60, M0 = 120, ef_construction = 480, ortho_count = 1,
    prune_overflow = 0), graph built by the batched GPU builder, 10k queries, k = 10, ef_search = 60, fp32
    rows and uint8 rows + fp32 re-score (`use_compression`).  Properties over all results; 64 queries
    bit-equal (ids, distances, RECORD_STATS' distance count) to the oracle's walk of the same index file;
    recall@10 against the exact brute-force answer is reported and must clear what iid rows allow."""
    torch = pytest.importorskip("torch")
    from expann_amd import AntitopoEngine, GpuBruteForceEngine
    n, d, m, k, ef = 1_000_000, 128, 10_000, 10, 60
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    base_t = torch.randn(n, d, device="cuda", generator=g).abs_().mul_(40).round_().clamp_(0, 255)
    g.manual_seed(4321)
    q_t = torch.randn(m, d, device="cuda", generator=g).abs_().mul_(40).round_().clamp_(0, 255)
    base, q = base_t.cpu().numpy(), q_t.cpu().numpy()
    eng = AntitopoEngine(60, 480, 1, 0, False, dim=d)
    eng.store_many_vectors_batched(base, False, 1024)
    eng.build()
    assert eng.size() == n
    idx = str(tmp_path / "c4_1m.index")
    eng.save_index(idx)
    engc = AntitopoEngine(60, 480, 1, 0, True, dim=d)
    engc.load_index(idx)
    # exact answer for the recall figure (src/basic_bench.h:116-121,143)
    bf = GpuBruteForceEngine(d, "l2")
    bf.set_base_device(base_t.data_ptr(), n, 0)
    gt = torch.empty(m, k, dtype=torch.int64, device="cuda")
    gd = torch.empty(m, k, dtype=torch.float32, device="cuda")
    bf.search_device(q_t.data_ptr(), m, k, gt.data_ptr(), gd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    gt_h = gt.cpu().numpy().view(np.uint64)
    bf.close()
    og = oracle.Graph(idx)
    sel = np.arange(0, m, m // 64)[:64]
    recalls = {}
    for comp, e in ((False, eng), (True, engc)):
        e.set_ef_search(ef)
        oids, od, odc = og.query_k(q[sel], k, ef, comp)
        before = int(e.param_list()["num_distcomps"])
        ids1, d1 = e.query_many(q[sel], k)      # the checked queries alone: RECORD_STATS' distance count too
        assert int(e.param_list()["num_distcomps"]) - before == int(odc.sum()), comp
        assert np.array_equal(ids1, oids) and np.array_equal(d1.view(np.uint32), od.view(np.uint32)), comp
        ids, dists = e.query_many(q, k)
        _properties(torch, torch.from_numpy(ids.view(np.int64)), torch.from_numpy(dists), n)
        assert np.array_equal(ids[sel], oids) and np.array_equal(dists[sel].view(np.uint32), od.view(np.uint32)), comp
        recalls[comp] = oracle.recall(ids, gt_h)
    print("C4 1M rows ef=60 recall@10: fp32 %.4f, uint8 %.4f" % (recalls[False], recalls[True]))
    assert recalls[False] > 0.45 and recalls[True] > 0.45, recalls
    eng.close()
    engc.close()
