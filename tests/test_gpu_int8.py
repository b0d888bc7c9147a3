"""GPU parity of the 8-bit paths: integer scores are exact, so ids and
distances must be bit-identical to the oracle for every metric, including the reference's own
wrapped int8 kernel (src/distance.h:29-53)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(base, metric, dtype):
    from expann_amd import GpuBruteForceEngine
    eng = GpuBruteForceEngine(base.shape[1], metric, dtype)
    eng.store_many_vectors(base)
    eng.build()
    return eng


def _check(oracle, eng, base, queries, k, ometric):
    ids, dists = eng.query_k_batch(queries, k)
    rids, rd = oracle.brute_force(base, queries, k, ometric, n_threads=8)
    assert np.array_equal(ids, rids)
    assert np.array_equal(dists.view(np.uint32), rd.view(np.uint32))


def _sift_like(rng, n, d):
    """SURVEY 8d's SIFT stand-in: clamp(round(|N(0,1)|*40), 0, 255)."""
    return np.clip(np.round(np.abs(rng.standard_normal((n, d))) * 40), 0, 255)


@pytest.mark.parametrize("n,d,m,k", [(5000, 128, 20, 10), (40000, 128, 7, 10), (3000, 64, 33, 5),
                                     (2000, 256, 4, 10), (20000, 128, 1, 100)])
def test_u8_compressed_l2(oracle, n, d, m, k):
    """dist2_compressed (src/antitopo_engine.h:38-61): fp32 query truncated to integers against
    uint8 rows.  Ties are frequent on integer data, so this also exercises the (score, id) rule."""
    rng = np.random.RandomState(n + d)
    base = _sift_like(rng, n, d).astype(np.uint8)
    queries = (_sift_like(rng, m, d) + rng.uniform(0, 0.99, size=(m, d))).astype(np.float32)
    queries = np.minimum(queries, np.float32(255.5))
    eng = _engine(base, "l2", "u8")
    _check(oracle, eng, base, queries, k, oracle.METRIC_L2_U8)
    for tq in (1, 4, 16):
        if d == 128:
            eng.set_option("query_tile", tq)
            _check(oracle, eng, base, queries, k, oracle.METRIC_L2_U8)
    eng.close()


def test_u8_rejects_queries_outside_8_bits(oracle):
    from expann_amd._lib import ExpannError
    base = np.zeros((2000, 128), np.uint8)
    eng = _engine(base, "l2", "u8")
    q = np.zeros((2, 128), np.float32)
    q[1, 5] = 300.0
    with pytest.raises(ExpannError):
        eng.query_k_batch(q, 3)
    q[1, 5] = -1.0
    with pytest.raises(ExpannError):
        eng.query_k_batch(q, 3)
    eng.close()


@pytest.mark.parametrize("metric,ometric", [("l2", "METRIC_L2_I8"), ("ip", "METRIC_IP_I8"),
                                            ("l2_i8_refcompat", "METRIC_L2_I8_REFCOMPAT")])
@pytest.mark.parametrize("n,d,m,k", [(6000, 128, 19, 10), (3000, 768, 9, 10), (30000, 64, 5, 10),
                                     (5000, 832, 6, 10), (5000, 960, 3, 10)])
def test_int8_metrics(oracle, metric, ometric, n, d, m, k):
    rng = np.random.RandomState(n * 3 + d)
    base = rng.randint(-127, 128, size=(n, d)).astype(np.int8)
    queries = rng.randint(-127, 128, size=(m, d)).astype(np.int8)
    eng = _engine(base, metric, "i8")
    _check(oracle, eng, base, queries, k, getattr(oracle, ometric))
    eng.close()


def test_int8_refcompat_differs_from_true_l2_like_the_reference(oracle):
    """On [0,127] data the reference kernel's ranking is NOT the L2 ranking; both
    are available and both match their oracle."""
    rng = np.random.RandomState(31)
    base = rng.randint(0, 128, size=(4000, 128)).astype(np.int8)
    queries = rng.randint(0, 128, size=(6, 128)).astype(np.int8)
    e1 = _engine(base, "l2", "i8")
    e2 = _engine(base, "l2_i8_refcompat", "i8")
    i1, _ = e1.query_k_batch(queries, 10)
    i2, d2 = e2.query_k_batch(queries, 10)
    assert not np.array_equal(i1, i2)
    rids, rd = oracle.brute_force(base, queries, 10, oracle.METRIC_L2_I8_REFCOMPAT)
    assert np.array_equal(i2, rids) and np.array_equal(d2, rd)
    e1.close()
    e2.close()


def test_score_ids_8bit(oracle):
    rng = np.random.RandomState(77)
    base = _sift_like(rng, 3000, 128).astype(np.uint8)
    q = (_sift_like(rng, 1, 128)[0] + 0.5).astype(np.float32)
    eng = _engine(base, "l2", "u8")
    ids = rng.randint(0, 3000, size=120).astype(np.uint64)
    cutoff = 150000.0
    kept, sc = eng.score_ids(q, ids, cutoff)
    okept, osc = oracle.filter_by_score(base, q, ids, cutoff, oracle.METRIC_L2_U8)
    assert np.array_equal(kept, okept) and np.array_equal(sc, osc)
    eng.close()
    b8 = rng.randint(-127, 128, size=(2000, 768)).astype(np.int8)
    q8 = rng.randint(-127, 128, size=768).astype(np.int8)
    eng = _engine(b8, "ip", "i8")
    ids = rng.randint(0, 2000, size=50).astype(np.uint64)
    kept, sc = eng.score_ids(q8, ids)
    okept, osc = oracle.filter_by_score(b8, q8, ids, float("inf"), oracle.METRIC_IP_I8)
    assert np.array_equal(kept, okept) and np.array_equal(sc, osc)
    eng.close()


@pytest.mark.parametrize("dtype,metric,ometric,d", [
    ("u8", "l2", "METRIC_L2_U8", 128), ("u8", "l2", "METRIC_L2_U8", 256),
    ("i8", "l2", "METRIC_L2_I8", 128), ("i8", "ip", "METRIC_IP_I8", 768),
    ("i8", "l2", "METRIC_L2_I8", 768), ("i8", "ip", "METRIC_IP_I8", 256),
])
@pytest.mark.parametrize("n,m,k", [(20000, 130, 10), (4099, 100, 17), (300, 129, 10)])
def test_gemm_form_int8_mfma(oracle, dtype, metric, ometric, d, n, m, k):
    """scan_kernel=2 forces the int8-MFMA GEMM-form filter (exact in integers, no re-rank)."""
    rng = np.random.RandomState(n + d + m)
    if dtype == "u8":
        base = _sift_like(rng, n, d).astype(np.uint8)
        queries = np.minimum(_sift_like(rng, m, d) + rng.uniform(0, 0.99, size=(m, d)),
                             255.5).astype(np.float32)
    else:
        base = rng.randint(-128, 128, size=(n, d)).astype(np.int8)
        queries = rng.randint(-128, 128, size=(m, d)).astype(np.int8)
    eng = _engine(base, metric, dtype)
    eng.set_option("scan_kernel", 2)
    eng.set_profiling(True)
    _check(oracle, eng, base, queries, k, getattr(oracle, ometric))
    prof = eng.get_profile()
    if n > 1024:
        assert prof["scan_kernel"].startswith("scan_gemm_i8"), prof["scan_kernel"]
    eng.close()


@pytest.mark.parametrize("dtype,metric,ometric,d", [
    ("u8", "l2", "METRIC_L2_U8", 128), ("u8", "l2", "METRIC_L2_U8", 256),
    ("i8", "l2", "METRIC_L2_I8", 128), ("i8", "ip", "METRIC_IP_I8", 128),
    ("i8", "l2", "METRIC_L2_I8", 256), ("i8", "ip", "METRIC_IP_I8", 256),
    ("i8", "ip", "METRIC_IP_I8", 768), ("i8", "l2", "METRIC_L2_I8", 768), ("u8", "l2", "METRIC_L2_U8", 768),
    # d = 832 / 960 (the reference's other builds): rows padded to 1024 bytes in the engine's copy
    ("u8", "l2", "METRIC_L2_U8", 832), ("i8", "ip", "METRIC_IP_I8", 832), ("i8", "l2", "METRIC_L2_I8", 960),
    ("u8", "l2", "METRIC_L2_U8", 960), ("i8", "ip", "METRIC_IP_I8", 960),
])
@pytest.mark.parametrize("n,m,k", [(40000, 130, 10), (70001, 300, 17), (65536, 97, 100)])
def test_gemm_form_int8_queues(oracle, dtype, metric, ometric, d, n, m, k):
    """scan_kernel=5: the int8-MFMA form in the fp16 kernel's geometry (sampled class-maxima pass
    for the threshold, per-wave hit queues; scan_gemm_i8q.hpp) -- exact integer scores, ties by
    row number, ragged last tile (the engine pads its own copy)."""
    rng = np.random.RandomState(n + d + m)
    if dtype == "u8":
        base = _sift_like(rng, n, d).astype(np.uint8)
        queries = np.minimum(_sift_like(rng, m, d) + rng.uniform(0, 0.99, size=(m, d)),
                             255.5).astype(np.float32)
    else:
        base = rng.randint(-128, 128, size=(n, d)).astype(np.int8)
        queries = rng.randint(-128, 128, size=(m, d)).astype(np.int8)
    eng = _engine(base, metric, dtype)
    eng.set_option("scan_kernel", 5)
    eng.set_profiling(True)
    _check(oracle, eng, base, queries, k, getattr(oracle, ometric))
    # (d >= 768: the 16x16x64 form scan_gemm_i8x of the same geometry, scan_gemm_i8x.hpp; d = 128 / 256:
    # scan_gemm_i8w, the 16x16x64 form in scan_gemm_f16x's step structure with hit logs)
    want = "scan_gemm_i8x" if d >= 768 else "scan_gemm_i8w"
    assert eng.get_profile()["scan_kernel"].startswith(want), eng.get_profile()["scan_kernel"]
    eng.close()


@pytest.mark.parametrize("opt", ["i8w", "i8x", "f16x"])
def test_removed_32x32_forms_are_refused_loudly(opt):
    """round 3 deleted the 32x32 MFMA kernels of rounds 1-2 (scan_gemm_i8q / scan_gemm_f16 / f16k): the options
    that selected them (i8w = 0, i8x = 0, f16x = 0) fail instead of silently running something else; = 1 is
    accepted (it is what runs)."""
    rng = np.random.RandomState(3)
    base = rng.randint(0, 256, size=(5000, 128)).astype(np.uint8)
    eng = _engine(base, "l2", "u8")
    eng.set_option(opt, 1)
    with pytest.raises(Exception) as ei:
        eng.set_option(opt, 0)
    assert "removed in round 3" in str(ei.value)
    eng.close()


def test_int8_queues_overflow_retries(oracle):
    """tiny candidate buffers: the queue form must notice the overflow, grow and repeat"""
    rng = np.random.RandomState(99)
    base = _sift_like(rng, 70001, 128).astype(np.uint8)
    queries = _sift_like(rng, 130, 128).astype(np.float32)
    eng = _engine(base, "l2", "u8")
    eng.set_option("cand_capacity", 64)
    eng.set_profiling(True)
    _check(oracle, eng, base, queries, 10, oracle.METRIC_L2_U8)
    prof = eng.get_profile()
    # (the retry stays on the hit-log kernel: the logs grow with the lists)
    assert prof["scan_kernel"].startswith("scan_gemm_i8w") and prof["retries"] >= 1
    eng.close()


@pytest.mark.parametrize("n,d,m,k,spread", [(20000, 128, 40, 10, 90), (4099, 64, 7, 17, 90),
                                             (3000, 128, 20, 10, 20000)])
def test_int16_rows_refcompat(oracle, n, d, m, k, spread):
    """int16 rows with the arithmetic of src/distance.h:14-27 (16-bit wrapping
    subtract and square, sign-extended sum).  spread 90: |a-b| <= 180, where it equals the true
    L2; spread 20000: differences far beyond 181, where the reference's value is garbage --
    and the GPU must reproduce exactly that garbage, ties by row number included."""
    rng = np.random.RandomState(n + d)
    base = rng.randint(-spread, spread + 1, size=(n, d)).astype(np.int16)
    queries = rng.randint(-spread, spread + 1, size=(m, d)).astype(np.int16)
    eng = _engine(base, "l2", "i16")
    _check(oracle, eng, base, queries, k, oracle.METRIC_L2_I16_REFCOMPAT)
    if spread <= 90:   # no wrap: also the true squared L2
        ids, dists = eng.query_k_batch(queries[:3], 1)
        want = ((base[ids[:, 0].astype(np.int64)].astype(np.int64) - queries[:3].astype(np.int64)) ** 2).sum(1)
        assert np.array_equal(dists[:, 0].astype(np.int64), want)
    ids = rng.randint(0, n, size=50).astype(np.uint64)
    kept, sc = eng.score_ids(queries[0], ids)
    okept, osc = oracle.filter_by_score(base, queries[0], ids, float("inf"), oracle.METRIC_L2_I16_REFCOMPAT)
    assert np.array_equal(kept, okept) and np.array_equal(sc, osc)
    eng.close()


def test_quantizer_builds_on_device(oracle):
    """quantizer_simple<uint8_t> (cast) and quantizer_ranged_q8 (affine int8) vs the oracle."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    from expann_amd import _lib
    L = _lib.load()
    rng = np.random.RandomState(12)
    x = np.clip(np.abs(rng.standard_normal((500, 128))) * 40, 0, 255.9).astype(np.float32)
    want = np.empty(x.size, np.uint8)
    oracle.lib().oracle_quantize_simple_u8(x.ctypes.data, x.size, want.ctypes.data)
    dx = torch.from_numpy(x).cuda()
    out = torch.empty(x.size, dtype=torch.uint8, device="cuda")
    assert L.expann_quantize_simple_u8_device(0, dx.data_ptr(), x.size, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    for scale in (1.0, 37.5, 1e-3):
        y = (rng.standard_normal((300, 64)) * scale).astype(np.float32)
        q8 = np.empty(y.size, np.int8)
        sf, off = C.c_float(), C.c_float()
        oracle.lib().oracle_quantize_ranged_q8(y.ctypes.data, 300, 64, q8.ctypes.data,
                                               C.byref(sf), C.byref(off))
        dy = torch.from_numpy(y).cuda()
        o8 = torch.empty(y.size, dtype=torch.int8, device="cuda")
        so = torch.empty(2, dtype=torch.float32, device="cuda")
        assert L.expann_quantize_ranged_q8_device(0, dy.data_ptr(), y.size, o8.data_ptr(),
                                                  so.data_ptr(), None) == 0
        assert np.array_equal(o8.cpu().numpy(), q8)
        assert so.cpu().numpy()[0] == np.float32(sf.value) and so.cpu().numpy()[1] == np.float32(off.value)
