"""Seeded random sweep over shapes, row types, metrics, data distributions and engine options:
every combination must return the oracle's ids and fp32 distances bit for bit.  The point is the
seams between the paths (direct scan / threshold ladder / sampled pass / hit queues / wave select /
overflow retries / uint8 shortcut), which single-purpose tests visit one at a time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(rng, kind, n, d):
    if kind == "gauss":
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == "clustered":
        c = rng.randint(8, 400)
        centres = rng.standard_normal((c, d)).astype(np.float32)
        per = (n + c - 1) // c
        return (np.repeat(centres, per, 0)[:n] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    if kind == "integers":                     # SIFT-like: fp32 rows holding 0..255
        return np.clip(np.round(np.abs(rng.standard_normal((n, d))) * 40), 0, 255).astype(np.float32)
    if kind == "dups":                         # many exact duplicates
        x = rng.standard_normal((n, d)).astype(np.float32)
        x[rng.randint(0, n, n // 3)] = x[rng.randint(0, n)]
        return x
    if kind == "scaled":                       # norms spread over two decades
        return (rng.standard_normal((n, d)) * 10.0 ** rng.uniform(-1, 1, size=(n, 1))).astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EXPANN_FUZZ_N", "40"))))
def test_random_configuration_matches_the_oracle(oracle, seed):
    from expann_amd import GpuBruteForceEngine
    rng = np.random.RandomState(1000 + seed)
    dtype = rng.choice(["f32", "f32", "f32", "u8", "i8"])
    d = int(rng.choice([64, 128, 128, 256, 512, 832, 960] if dtype == "f32" else [128, 128, 256, 768, 832, 960]))
    n = int(rng.choice([700, 5000, 20000, 40000, 70001, 131072]))
    m = int(rng.choice([1, 3, 5, 8, 23, 64, 97, 130, 300]))
    k = int(rng.choice([1, 5, 10, 10, 17, 64, 100]))
    metric = "l2"
    if dtype == "f32":
        metric = str(rng.choice(["l2", "l2", "ip"]))
        kind = str(rng.choice(["gauss", "clustered", "integers", "dups", "scaled"]))
        base = _data(rng, kind, n, d)
        queries = _data(rng, kind if kind != "dups" else "gauss", m, d)
        if kind == "dups":
            queries[0] = base[0]
        ometric = oracle.METRIC_L2_F32 if metric == "l2" else oracle.METRIC_IP_F32
    elif dtype == "u8":
        base = np.clip(np.round(np.abs(rng.standard_normal((n, d))) * 40), 0, 255).astype(np.uint8)
        queries = np.clip(np.round(np.abs(rng.standard_normal((m, d))) * 40), 0, 255).astype(np.float32)
        ometric = oracle.METRIC_L2_U8
    else:
        metric = str(rng.choice(["l2", "ip"]))
        base = rng.randint(-128, 128, size=(n, d)).astype(np.int8)
        queries = rng.randint(-128, 128, size=(m, d)).astype(np.int8)
        ometric = oracle.METRIC_L2_I8 if metric == "l2" else oracle.METRIC_IP_I8
    eng = GpuBruteForceEngine(d, metric, dtype)
    eng.store_many_vectors(base)
    eng.build()
    if rng.rand() < 0.25:
        eng.set_option("cand_capacity", int(rng.choice([64, 256, 1024])))
    if rng.rand() < 0.2:
        eng.set_option("sample_frac", int(rng.choice([4, 32])))
    ids, dists = eng.query_k_batch(queries, k)
    rids, rd = oracle.brute_force(base, queries, k, ometric, n_threads=8)
    eng.close()
    assert np.array_equal(ids, rids), (dtype, metric, n, d, m, k)
    assert np.array_equal(dists.view(np.uint32), rd.view(np.uint32)), (dtype, metric, n, d, m, k)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EXPANN_FUZZ_MFMA_N", "24"))))
def test_random_mfma_configuration_matches_the_oracle(oracle, seed):
    """The same sweep with every case large enough for the matrix-core filters (>= 65 536 rows, >= 97
    queries): fp16 forms at d = 64 ... 960, int8 forms at d = 128 ... 960, their sampled passes, hit
    logs / queues, overflow retries with tiny lists; except on massive-tie data the kernel that ran must
    be one of the GEMM forms."""
    from expann_amd import GpuBruteForceEngine
    rng = np.random.RandomState(50_000 + seed)
    dtype = rng.choice(["f32", "f32", "u8", "i8"])
    d = int(rng.choice([64, 128, 256, 512, 768, 832, 960] if dtype == "f32" else [128, 256, 768, 832, 960]))
    n = int(rng.choice([70001, 131072, 200000]))
    m = int(rng.choice([97, 130, 300, 700]))
    k = int(rng.choice([1, 10, 10, 17, 64, 100]))
    metric = "l2"
    if dtype == "f32":
        metric = str(rng.choice(["l2", "l2", "ip"]))
        kind = str(rng.choice(["gauss", "gauss", "clustered", "dups", "scaled"]))
        base = _data(rng, kind, n, d)
        queries = _data(rng, kind if kind != "dups" else "gauss", m, d)
        if kind == "dups":
            queries[0] = base[0]
        ometric = oracle.METRIC_L2_F32 if metric == "l2" else oracle.METRIC_IP_F32
    elif dtype == "u8":
        base = np.clip(np.round(np.abs(rng.standard_normal((n, d))) * 40), 0, 255).astype(np.uint8)
        queries = np.clip(np.round(np.abs(rng.standard_normal((m, d))) * 40), 0, 255).astype(np.float32)
        ometric = oracle.METRIC_L2_U8
    else:
        metric = str(rng.choice(["l2", "ip"]))
        base = rng.randint(-128, 128, size=(n, d)).astype(np.int8)
        queries = rng.randint(-128, 128, size=(m, d)).astype(np.int8)
        ometric = oracle.METRIC_L2_I8 if metric == "l2" else oracle.METRIC_IP_I8
    eng = GpuBruteForceEngine(d, metric, dtype)
    eng.store_many_vectors(base)
    eng.build()
    eng.set_profiling(True)
    if rng.rand() < 0.2:
        eng.set_option("cand_capacity", int(rng.choice([256, 1024])))
    if rng.rand() < 0.2:
        eng.set_option("sample_frac", int(rng.choice([4, 32])))
    ids, dists = eng.query_k_batch(queries, k)
    kernel = eng.get_profile()["scan_kernel"]
    rids, rd = oracle.brute_force(base, queries, k, ometric, n_threads=16)
    eng.close()
    assert np.array_equal(ids, rids), (dtype, metric, n, d, m, k, kernel)
    assert np.array_equal(dists.view(np.uint32), rd.view(np.uint32)), (dtype, metric, n, d, m, k, kernel)
    # (massive ties -- "dups" -- may end on the exact direct kernel, which breaks them by row number)
    if not (dtype == "f32" and kind == "dups"):
        assert kernel.startswith("scan_gemm_"), (dtype, metric, n, d, m, k, kernel)
