"""GPU tests at BASELINE.json's full single-GPU size (C2: 1M x d128 fp32, 10k queries, k=10):
a sampled bit-exact check against the oracle plus size-independent properties (sortedness,
no duplicate ids, idempotence, shard-and-merge == unsharded)."""
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
