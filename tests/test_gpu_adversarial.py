"""Adversarial soundness of the conservative fp16 filter (scan_gemm_f16.hpp / scan_gemm_f16k.hpp).

The filter may only DROP a row when its reference-order score is certainly above the threshold;
its slack  eps_d (|q|^2 + |b|^2) + abs (|q| + |b|)  is 12.5 % (d <= 256) above the worst case of
the fp16 input rounding, and that worst case needs (1) every component of q and of b rounded in
the SAME direction by the full half-ulp, (2) |q_i| = |b_i| in every component (q ~ b: the
expanded form cancels catastrophically) and (3) a threshold with no room of its own.  Random data
never comes near it (errors grow like sqrt(d), not d).  These tests construct it:

  * every component is  +-2^e (1 + j 2^-10 + 2^-11 -+ 2^-23), j small: one fp32 ulp beside the
    midpoint of two fp16 neighbours, so fp16 rounding moves ALL of them down (or all up) by
    ~2^-11 relative;
  * the true neighbours of a query are the query itself with a few components doubled (same
    mantissa, same rounding direction, score = sum of those q_i^2 << |q|^2 + |b|^2);
  * they sit in rows that every threshold sample contains, so with the threshold ladder
    (sample_pass = 0) tau is EXACTLY the k-th neighbour's reference-order score: that row has to
    pass `estimate <= tau` with nothing but the slack between them.

`bite` (host arithmetic, float64) is the fraction of the slack the construction really consumes;
the asserts on it keep the test honest.  Ids and distance bits must equal the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, M, K = 65536, 40, 10


def _adv(rng, shape, direction):
    """fp32 values one ulp beside an fp16 rounding midpoint; direction -1: fp16 rounds toward zero,
    +1: away from zero, 0: half / half."""
    e = rng.randint(-3, 1, size=shape)
    j = rng.randint(0, 6, size=shape)
    sgn = rng.choice([-1.0, 1.0], size=shape)
    if direction == 0:
        dirs = rng.choice([-1.0, 1.0], size=shape)
    else:
        dirs = np.full(shape, float(direction))
    mant = 1.0 + j * 2.0 ** -10 + 2.0 ** -11 + dirs * 2.0 ** -23
    v = (sgn * mant * 2.0 ** e).astype(np.float32)
    assert np.array_equal(v.astype(np.float64), sgn * mant * 2.0 ** e)      # exactly representable
    return v


def _build(d, direction, seed):
    rng = np.random.RandomState(seed)
    q = _adv(rng, (M, d), direction)
    qn = float(np.sqrt((q[0].astype(np.float64) ** 2).sum()))
    base = (rng.standard_normal((N, d)) * qn / np.sqrt(d)).astype(np.float32)    # far rows, same norm
    fam = np.zeros((M, K), dtype=np.int64)
    for t in range(M):
        order = np.argsort(np.abs(q[t]), kind="stable")      # smallest components first
        for jn in range(K):
            row = 1024 * t + jn                               # inside every threshold sample
            b = q[t].copy()
            b[order[:jn + 1]] *= 2.0                          # same mantissa, same rounding direction
            base[row] = b
            fam[t, jn] = row
    base[1024 * M] = q[0]                                     # an exact duplicate of a query (score 0)
    return base, q, fam


def _bite(base, q, fam, d):
    """largest fraction of the L2 slack consumed by a family row (float64 on the exact values)"""
    from math import sqrt
    eps = 1.125 * 2.0 ** -10 if d <= 256 else 2.0 ** -10 + (4.0 + 4.25 * d + 256.0) * 2.0 ** -24
    maxabs = max(np.abs(base).max(), 1e-30)
    s = 2.0 ** np.floor(np.log2(32768.0 / maxabs))
    worst = 0.0
    for t in range(M):
        qt = q[t].astype(np.float64)
        q16 = (q[t] * np.float32(s)).astype(np.float16).astype(np.float64)
        for row in fam[t]:
            b = base[row].astype(np.float64)
            b16 = (base[row] * np.float32(s)).astype(np.float16).astype(np.float64)
            nq, nb = (qt ** 2).sum(), (b ** 2).sum()
            true = ((qt - b) ** 2).sum()
            raw = (nq + nb - 2.0 * (q16 * b16).sum() / s ** 2) - true
            slack = eps * (nq + nb) + 2.0 ** -24 / s * sqrt(d) * (sqrt(nq) + sqrt(nb))
            worst = max(worst, raw / slack)
    return worst


@pytest.mark.parametrize("d", [128, 512, 960])
@pytest.mark.parametrize("direction", [-1, 1, 0])
def test_same_signed_rounding_with_cancellation_and_exact_thresholds(oracle, d, direction):
    from expann_amd import GpuBruteForceEngine
    base, q, fam = _build(d, direction, 1000 + d + direction)
    if direction == -1:
        bite = _bite(base, q, fam, d)
        # theory: 1 / 1.125 = 0.889 (d <= 256), 0.87 at d = 512, 0.79 at d = 960, reached up to the
        # j-dependent mantissas and the far rows' share of the scale
        assert bite > (0.80 if d <= 512 else 0.72), bite
    for metric, om in (("l2", oracle.METRIC_L2_F32), ("ip", oracle.METRIC_IP_F32)):
        ref = oracle.brute_force(base, q, K, om, n_threads=16)
        if metric == "l2":   # the construction is what it claims: the family rows ARE the answer
            assert np.array_equal(np.sort(ref[0][1:], axis=1), np.sort(fam[1:].astype(np.uint64), axis=1))
        for opts in ({}, {"sample_pass": 0, "scan_kernel": 4}, {"scan_kernel": 4, "sample_frac": 4}):
            eng = GpuBruteForceEngine(d, metric)
            eng.store_many_vectors(base)
            eng.build()
            for name, val in opts.items():
                eng.set_option(name, val)
            eng.set_profiling(True)
            ids, dists = eng.query_k_batch(q, K)
            prof = eng.get_profile()
            eng.close()
            assert prof["scan_kernel"].startswith("scan_gemm_f16"), prof["scan_kernel"]   # the filter under test ran
            assert np.array_equal(ids, ref[0]), (metric, opts)
            assert np.array_equal(dists.view(np.uint32), ref[1].view(np.uint32)), (metric, opts)


def test_bf16_split_filter_same_signed_rounding(oracle):
    """the 3-term bf16 split (scan_kernel = 3): components one ulp beside a bf16 rounding midpoint in
    the hi part, same construction otherwise"""
    from expann_amd import GpuBruteForceEngine
    d = 128
    rng = np.random.RandomState(4242)
    e = rng.randint(-3, 1, size=(M, d))
    sgn = rng.choice([-1.0, 1.0], size=(M, d))
    # hi = bf16(x) keeps 8 bits: midpoint at 2^-8; lo = bf16(x - hi) keeps 8 more: put the value one
    # fp32 ulp below a midpoint of the SECOND split as well
    mant = 1.0 + 2.0 ** -8 - 2.0 ** -16 - 2.0 ** -23
    q = (sgn * mant * 2.0 ** e).astype(np.float32)
    qn = float(np.sqrt((q[0].astype(np.float64) ** 2).sum()))
    base = (rng.standard_normal((N, d)) * qn / np.sqrt(d)).astype(np.float32)
    for t in range(M):
        order = np.argsort(np.abs(q[t]), kind="stable")
        for jn in range(K):
            b = q[t].copy()
            b[order[:jn + 1]] *= 2.0
            base[1024 * t + jn] = b
    ref = oracle.brute_force(base, q, K, oracle.METRIC_L2_F32, n_threads=16)
    eng = GpuBruteForceEngine(d, "l2")
    eng.store_many_vectors(base)
    eng.build()
    eng.set_option("scan_kernel", 3)
    ids, dists = eng.query_k_batch(q, K)
    eng.close()
    assert np.array_equal(ids, ref[0]) and np.array_equal(dists.view(np.uint32), ref[1].view(np.uint32))
