"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against
the CPU oracle on the same seeded inputs.  fp32 results are required to be BIT-EXACT (ids and
distance bits): the kernels reproduce the reference's 16-lane FMA order, so the
north_star's 1e-4 relative tolerance is met with zero slack."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAD = np.uint64(2 ** 64 - 1)


@pytest.fixture(scope="module")
def gpu():
    from expann_amd import _lib
    L = _lib.load()
    assert L.expann_device_count() >= 1, "these tests need a HIP device"
    return L


def _engine(base, metric="l2"):
    from expann_amd import GpuBruteForceEngine
    eng = GpuBruteForceEngine(base.shape[1], metric)
    eng.store_many_vectors(base)
    eng.build()
    return eng


def _check(oracle, base, queries, k, metric="l2", eng=None):
    own = eng is None
    eng = eng or _engine(base, metric)
    ids, dists = eng.query_k_batch(queries, k)
    om = oracle.METRIC_L2_F32 if metric == "l2" else oracle.METRIC_IP_F32
    rids, rd = oracle.brute_force(base, queries, k, om, n_threads=8)
    assert np.array_equal(ids, rids)
    assert np.array_equal(dists.view(np.uint32), rd.view(np.uint32))
    if own:
        eng.close()
    return ids, dists


@pytest.mark.parametrize("n,d,m,k", [
    (10000, 64, 100, 10),     # BASELINE config C1
    (1000, 128, 7, 1),
    (20000, 128, 33, 10),
    (5000, 128, 16, 100),
    (50000, 128, 50, 10),
    (3000, 256, 9, 10),
    (2000, 960, 5, 10),
    (4099, 128, 3, 17),       # n not a multiple of 16
])
def test_l2_bit_exact_vs_oracle(gpu, oracle, n, d, m, k):
    rng = np.random.RandomState(n + d + m + k)
    base = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((m, d)).astype(np.float32)
    _check(oracle, base, queries, k)


@pytest.mark.parametrize("n,d,m,k", [(10000, 64, 20, 10), (30000, 128, 17, 10)])
def test_ip_bit_exact_vs_oracle(gpu, oracle, n, d, m, k):
    rng = np.random.RandomState(99 + n)
    base = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((m, d)).astype(np.float32)
    _check(oracle, base, queries, k, metric="ip")


def test_small_and_ragged_indexes(gpu, oracle):
    rng = np.random.RandomState(5)
    for n in (1, 3, 15, 16, 17, 63, 1023, 1024, 1025):
        base = rng.standard_normal((n, 128)).astype(np.float32)
        queries = rng.standard_normal((5, 128)).astype(np.float32)
        ids, dists = _check(oracle, base, queries, 10)
        if n < 10:  # k > n: min(k, n) results, padded (brute_force_engine.h:39-45 returns n ids)
            assert np.all(ids[:, n:] == PAD) and np.all(np.isinf(dists[:, n:]))
            eng = _engine(base)
            assert len(eng.query_k(queries[0], 10)) == n
            eng.close()


def test_duplicate_rows_tie_break_by_lower_id(gpu, oracle):
    """Exact duplicates give equal distances: the reference keeps the lower ids
    (src/brute_force_engine.h:33-37)."""
    rng = np.random.RandomState(8)
    uniq = rng.standard_normal((700, 128)).astype(np.float32)
    base = np.concatenate([uniq, uniq, uniq[:300]], 0)   # every row 2-3 times
    queries = np.concatenate([rng.standard_normal((6, 128)).astype(np.float32), uniq[:4]], 0)
    ids, dists = _check(oracle, base, queries, 10)
    assert dists[6, 0] == 0.0 and ids[6, 0] == 0 and ids[6, 1] == 700 and ids[6, 2] == 1400
    # all rows identical: ids 0..k-1
    same = np.repeat(uniq[:1], 5000, 0)
    ids2, _ = _check(oracle, same, queries[:3], 10)
    assert np.array_equal(ids2[0], np.arange(10, dtype=np.uint64))


@pytest.mark.parametrize("m", [3, 130])
def test_massive_exact_ties_do_not_overflow(gpu, oracle, m):
    """40 000 identical rows (more than any candidate buffer holds) plus a few distinct ones: the
    threshold test is lexicographic on (score, row), so ties pass only up to the threshold's row;
    the GEMM forms (m = 130) cannot break exact ties and fall back to the direct scan."""
    rng = np.random.RandomState(55)
    one = rng.standard_normal((1, 128)).astype(np.float32)
    base = np.concatenate([rng.standard_normal((500, 128)).astype(np.float32),
                           np.repeat(one, 40000, 0),
                           rng.standard_normal((700, 128)).astype(np.float32)], 0)
    queries = np.concatenate([one + np.float32(0.01), rng.standard_normal((m - 1, 128)).astype(np.float32)], 0)
    ids, dists = _check(oracle, base, queries, 10)
    assert np.array_equal(ids[0], np.arange(500, 510, dtype=np.uint64))


def test_every_query_tile_gives_the_same_answer(gpu, oracle):
    rng = np.random.RandomState(21)
    base = rng.standard_normal((30000, 128)).astype(np.float32)
    queries = rng.standard_normal((37, 128)).astype(np.float32)
    eng = _engine(base)
    for tq in (1, 2, 4, 8, 16, 0):
        eng.set_option("query_tile", tq)
        _check(oracle, base, queries, 10, eng=eng)
    eng.close()


@pytest.mark.parametrize("n,d,m,k", [
    (20000, 128, 130, 10),
    (5000, 64, 200, 10),
    (4099, 128, 100, 17),     # ragged last tile
    (1000, 128, 97, 1),
    (70000, 128, 300, 100),
    (300, 128, 129, 10),      # fewer rows than one 128-row tile pair
])
@pytest.mark.parametrize("kernel,kname", [(2, "scan_gemm_f32"), (3, "scan_gemm_bf16x3")])
def test_gemm_form_scan_bit_exact_vs_oracle(gpu, oracle, n, d, m, k, kernel, kname):
    """scan_kernel=2/3 force the MFMA GEMM-form candidate filters (fp32-input MFMA, bf16 MFMA
    with the 3-term split) + exact re-rank in the select kernel: the final ids and distances
    must still be bit-identical to the oracle."""
    rng = np.random.RandomState(n * 7 + m)
    base = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((m, d)).astype(np.float32)
    eng = _engine(base)
    eng.set_option("scan_kernel", kernel)
    eng.set_profiling(True)
    _check(oracle, base, queries, k, eng=eng)
    prof = eng.get_profile()
    if n > 1024:
        assert prof["scan_kernel"].startswith(kname)
    eng.close()


@pytest.mark.parametrize("kernel", [2, 3])
def test_gemm_form_scan_near_duplicates_and_ties(gpu, oracle, kernel):
    """The expanded form ||b||^2 - 2q.b + ||q||^2 cancels catastrophically for near-duplicate
    vectors (SURVEY 7 'hard parts'); the slack + exact re-rank must keep the result exact."""
    rng = np.random.RandomState(42)
    uniq = (rng.standard_normal((3000, 128)) * 10).astype(np.float32)
    near = uniq + (rng.standard_normal(uniq.shape) * 1e-4).astype(np.float32)
    base = np.concatenate([uniq, near, uniq[:500]], 0)
    queries = np.concatenate([uniq[:100] + np.float32(1e-5), uniq[100:140]], 0)
    eng = _engine(base)
    eng.set_option("scan_kernel", kernel)
    _check(oracle, base, queries, 10, eng=eng)
    # wide dynamic range (values from 1e-3 to 1e3 in one vector) stresses the bf16 split
    wide = (rng.standard_normal((6000, 128)) * 10.0 ** rng.uniform(-3, 3, size=(6000, 128))).astype(np.float32)
    wq = (rng.standard_normal((128, 128)) * 10.0 ** rng.uniform(-3, 3, size=(128, 128))).astype(np.float32)
    eng2 = _engine(wide)
    eng2.set_option("scan_kernel", kernel)
    _check(oracle, wide, wq, 10, eng=eng2)
    eng2.close()
    eng.close()


def test_query_k_single_and_idempotent(gpu, oracle):
    rng = np.random.RandomState(3)
    base = rng.standard_normal((12345, 128)).astype(np.float32)
    q = rng.standard_normal((4, 128)).astype(np.float32)
    eng = _engine(base)
    rids, _ = oracle.brute_force(base, q, 10)
    for i in range(4):
        assert eng.query_k(q[i], 10) == [int(x) for x in rids[i]]
    a = eng.query_k_batch(q, 10)
    b = eng.query_k_batch(q, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    eng.close()


def test_candidate_overflow_retries_are_transparent(gpu, oracle):
    """Force tiny candidate buffers: the engine must retry with larger ones, not return a
    wrong answer."""
    rng = np.random.RandomState(13)
    base = rng.standard_normal((40000, 128)).astype(np.float32)
    queries = rng.standard_normal((8, 128)).astype(np.float32)
    eng = _engine(base)
    eng.set_option("cand_capacity", 64)
    eng.set_option("sample_ratio", 1024)
    _check(oracle, base, queries, 10, eng=eng)
    eng.close()


def test_score_ids_matches_filter_by_score(gpu, oracle):
    rng = np.random.RandomState(17)
    base = rng.standard_normal((5000, 128)).astype(np.float32)
    q = rng.standard_normal(128).astype(np.float32)
    eng = _engine(base)
    for n_ids in (1, 5, 120, 1000):
        ids = rng.randint(0, 5000, size=n_ids).astype(np.uint64)
        all_ids, all_sc = eng.score_ids(q, ids)
        want = np.array([oracle.l2_f32(q, base[int(i)]) for i in ids], np.float32)
        assert np.array_equal(all_ids, ids)
        assert np.array_equal(all_sc.view(np.uint32), want.view(np.uint32))
        cutoff = float(np.median(want))
        kept, ksc = eng.score_ids(q, ids, cutoff)
        okept, oksc = oracle.filter_by_score(base, q, ids, cutoff)
        assert np.array_equal(kept, okept) and np.array_equal(ksc, oksc)
    eng.close()


def test_error_paths(gpu):
    from expann_amd import GpuBruteForceEngine
    from expann_amd._lib import ExpannError
    eng = GpuBruteForceEngine(128)
    with pytest.raises(ExpannError):      # build() on an empty index (reference asserts)
        eng.build()
    with pytest.raises(ExpannError):      # search before build
        eng.query_k_batch(np.zeros((1, 128), np.float32), 1)
    eng.store_many_vectors(np.zeros((4, 128), np.float32))
    eng.build()
    with pytest.raises(ExpannError):      # k == 0
        eng.query_k_batch(np.zeros((1, 128), np.float32), 0)
    with pytest.raises(ExpannError):      # id out of range
        eng.score_ids(np.zeros(128, np.float32), np.array([9], np.uint64))
    eng.close()
