"""Graph path (antitopo_engine): CPU tests of the host-side builder + the reference's index
format, and GPU parity of the traversal (expann_graph_search) against the oracle's restatement
of src/antitopo_engine.h:853-928 on the same index file: identical ids, distances and
distance-evaluation counts for every query, ef_search and both bottom-layer variants."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "expann_amd", "host", "expann_graph_tool")


def _tool(*args, cwd=None):
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(TOOL)])
    out = subprocess.run([TOOL] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd,
                         timeout=900)
    assert out.returncode == 0, out.stderr
    return [json.loads(x) for x in out.stdout.strip().splitlines()]


def test_builder_index_format_roundtrip_and_graph_quality(tmp_path, oracle):
    """Build on the CPU (no GPU needed), write the reference's index layout, read it back and
    rewrite it byte-identically; the oracle's search over it must behave like a graph index
    (recall grows with ef_search and reaches the exact answer on a small set)."""
    idx, idx2, qf = tmp_path / "g.index", tmp_path / "g2.index", tmp_path / "g.queries"
    lines = _tool("--n", 800, "--m", 40, "--d", 64, "--M", 12, "--ef_construction", 60,
                  "--data", "gauss", "--build-only", 1, "--index", idx, "--queries", qf)
    assert lines[0]["phase"] == "build" and lines[0]["n"] == 800 and lines[0]["max_layer"] >= 2
    _tool("--n", 0, "--m", 1, "--d", 64, "--build-only", 1, "--read-index", 1, "--index", idx,
          "--rewrite", idx2)
    assert open(idx, "rb").read() == open(idx2, "rb").read()
    g = oracle.Graph(str(idx))
    assert (g.n, g.dim) == (800, 64)
    q = np.fromfile(qf, dtype=np.float32).reshape(-1, 64)
    gt, _ = oracle.brute_force(g.vectors(), q, 10)
    recalls = []
    for ef in (10, 40, 200):
        ids, dists, dc = g.query_k(q, 10, ef)
        recalls.append(oracle.recall(ids, gt))
        assert np.all(np.diff(dists, axis=1) >= 0) and np.all(dc > 0)
    assert recalls[0] < recalls[1] <= recalls[2] and recalls[2] > 0.99


def test_strided_view_of_the_graph_is_lossless(tmp_path):
    """antitopo_index::to_strided / from_strided (the fixed-stride adjacency arrays the batched GPU
    builder works on) reproduce the index byte for byte -- CPU only."""
    idx, idx2 = tmp_path / "g.index", tmp_path / "g2.index"
    _tool("--n", 600, "--m", 1, "--d", 64, "--M", 8, "--ef_construction", 40, "--data", "gauss",
          "--build-only", 1, "--index", idx)
    _tool("--n", 0, "--m", 1, "--d", 64, "--build-only", 1, "--read-index", 1, "--index", idx,
          "--strided-roundtrip", 1, "--rewrite", idx2)
    assert open(idx, "rb").read() == open(idx2, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("data,d", [("sift", 128), ("gauss", 64), ("sift", 960), ("sift", 832), ("sift", 512), ("sift", 768),
                                    ("sift", 256)])
def test_gpu_traversal_matches_oracle(tmp_path, oracle, data, d):
    idx, qf, rf = tmp_path / "g.index", tmp_path / "g.queries", tmp_path / "g.results"
    n, m, k = 1500, 64, 10
    efs = [10, 25, 60]
    lines = _tool("--n", n, "--m", m, "--d", d, "--k", k, "--M", 16, "--ef_construction", 80,
                  "--data", data, "--index", idx, "--queries", qf, "--results", rf,
                  "--ef", ",".join(map(str, efs)))
    assert sum(1 for x in lines if x["phase"] == "query") == 2 * len(efs)
    g = oracle.Graph(str(idx))
    q = np.fromfile(qf, dtype=np.float32).reshape(m, d)
    raw = open(rf, "rb").read()
    pos = 0
    for comp in (0, 1):
        for ef in efs:
            ids = np.frombuffer(raw, np.uint64, m * k, pos).reshape(m, k); pos += m * k * 8
            dists = np.frombuffer(raw, np.float32, m * k, pos).reshape(m, k); pos += m * k * 4
            dc = np.frombuffer(raw, np.uint32, m, pos); pos += m * 4
            if comp and data == "gauss":
                continue  # uint8 truncation of N(0,1) data is meaningless (and negative -> UB)
            oids, odists, odc = g.query_k(q, k, ef, bool(comp))
            assert np.array_equal(ids, oids), (comp, ef)
            assert np.array_equal(dists.view(np.uint32), odists.view(np.uint32)), (comp, ef)
            assert np.array_equal(dc.astype(np.uint64), odc), (comp, ef)


def _sift_like(rng, n, d, frac=False):
    x = np.clip(np.round(np.abs(rng.standard_normal((n, d))) * 40.0), 0, 255).astype(np.float32)
    if frac:  # fractional parts: the uint8 path truncates the query (antitopo_engine.h:726-737)
        x = np.minimum(255.5, x + rng.uniform(0, 0.99, size=x.shape)).astype(np.float32)
    return x


@pytest.mark.gpu
@pytest.mark.parametrize("prune_overflow", [0, 1])
def test_c4_config_traversal_matches_oracle(tmp_path, oracle, prune_overflow):
    """The reference's own sweep configuration (src/bench_runner.h:133-162): M = 60, M0 = 120,
    ef_construction = 480, prune_overflow in {0, 1}, both compression modes, ef_search = k * {1..6}
    -- on a 3000-row SIFT-like index (the serial host build is the reference's, ~10 ms per row).
    Bottom-layer lists longer than one wavefront (deg > 64, graph_search.hpp's neighbour loop)
    occur here for the first time under the checker."""
    from graph_helpers import build_engines, check_against_oracle, read_index_degrees
    rng = np.random.RandomState(60 + prune_overflow)
    n, d, m, k = 3000, 128, 64, 10
    base = _sift_like(rng, n, d)
    q = _sift_like(rng, m, d, frac=True)
    engs, idx = build_engines(base, tmp_path, M=60, ef_construction=480, prune_overflow=prune_overflow)
    hdr, deg0 = read_index_degrees(idx)
    assert (hdr["M"], hdr["M0"], hdr["ef_construction"], hdr["prune_overflow"]) == (60, 120, 480, prune_overflow)
    assert deg0.max() > 64 and deg0.max() <= 120, deg0.max()
    g = check_against_oracle(oracle, engs, idx, q, k, efs=[k * mult for mult in (1, 2, 3, 4, 5, 6)])
    # and it is a usable index: recall@10 at ef = 60 against the exact answer
    gt, _ = oracle.brute_force(g.vectors(), q, k)
    engs[False].set_ef_search(60)
    ids, _ = engs[False].query_many(q, k)
    assert oracle.recall(ids, gt) > 0.9
    for e in engs.values():
        e.close()


@pytest.mark.gpu
def test_python_module_surface(tmp_path, oracle):
    """The pyrunner.cpp surface (AntitopoEngine): zero-padding to the engine dimension,
    take_norms, sticky ef_search, index save / load round trip; results checked against the oracle
    on the saved index."""
    from expann_amd import AntitopoEngine
    rng = np.random.RandomState(9)
    base = rng.standard_normal((1200, 100)).astype(np.float32)     # 100 dims -> padded to 128
    q = rng.standard_normal((20, 100)).astype(np.float32)
    eng = AntitopoEngine(16, 80, 1, 0, False)
    eng.store_many_vectors(base[:600], True)
    for v in base[600:610]:
        eng.store_vector(v / np.linalg.norm(v))
    eng.store_many_vectors(base[610:], True)
    assert eng.dim == 128 and eng.size() == 1200
    eng.build()
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    ids10 = [eng.query_k_numpy(x, 10) for x in qn]                  # ef_search := 10, sticky
    eng.set_ef_search(80)
    ids80, d80 = eng.query_many(qn, 10)
    idx = tmp_path / "py.index"
    eng.save_index(idx)
    g = oracle.Graph(str(idx))
    qp = np.zeros((20, 128), np.float32)
    qp[:, :100] = qn
    o10, _, _ = g.query_k(qp, 10, 10)
    o80, od80, _ = g.query_k(qp, 10, 80)
    assert [[int(x) for x in r] for r in o10] == ids10
    assert np.array_equal(ids80, o80) and np.array_equal(d80.view(np.uint32), od80.view(np.uint32))
    eng2 = AntitopoEngine(16, 80, 1, 0, False, dim=128)
    eng2.load_index(idx)
    eng2.set_ef_search(80)
    ids_b, _ = eng2.query_many(qn, 10)
    assert np.array_equal(ids_b, ids80)
    assert "M0" in eng.param_list() and int(eng.param_list()["num_distcomps"]) > 0
    eng.close()
    eng2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,efc,po", [(16, 100, 0), (60, 480, 1)])
def test_batched_gpu_builder_makes_a_valid_graph_of_the_same_quality(tmp_path, oracle, M, efc, po):
    """expann_graph_build_batched (csrc/graph_build.hpp) against the serial host restatement on the
    same rows: structural invariants of the index it writes (degrees within M / M0, no self loops, no
    duplicate edges, edge lengths = the reference-order distance of their endpoints bit for bit,
    every vertex reachable on layer 0), recall of the oracle's walk within a few points of the serial
    graph's, and the GPU traversal of the batched graph identical to the oracle's walk on it."""
    from graph_helpers import read_index_edges
    n, d, m, k = 12000, 128, 200, 10
    common = ["--n", n, "--m", m, "--d", d, "--k", k, "--M", M, "--ef_construction", efc, "--prune_overflow", po,
              "--data", "sift", "--build-only", 1]
    idx_b, idx_s, qf = tmp_path / "b.index", tmp_path / "s.index", tmp_path / "q.bin"
    lb = _tool(*common, "--batched", 1024, "--index", idx_b, "--queries", qf)
    assert lb[0]["builder"] == "batched gpu" and lb[0]["batches"] >= 10 and lb[0]["n"] == n
    assert lb[0]["dropped_reverse_edges"] == 0
    hdr, layers = read_index_edges(str(idx_b))
    gb = oracle.Graph(str(idx_b))
    vec = gb.vectors()
    cap = {0: 2 * M}
    seen0 = np.zeros(n, dtype=bool)
    for v, per_layer in enumerate(layers):
        for l, (ids, ds) in enumerate(per_layer):
            assert len(ids) <= cap.get(l, M), (v, l, len(ids))
            assert v not in ids and len(set(ids.tolist())) == len(ids)
            assert ids.max(initial=0) < n
            if l == 0:
                seen0[ids] = True
    # (the pruning heuristic may take a vertex's last incoming edge away, in the serial builder too)
    assert seen0.mean() > 0.98, "too many vertices without an incoming layer-0 edge"
    rng = np.random.RandomState(1)
    for v in rng.randint(0, n, 200):
        ids, ds = layers[v][0]
        ref = np.array([oracle.l2_f32(vec[v], vec[j]) for j in ids], dtype=np.float32)
        assert np.array_equal(ref.view(np.uint32), ds.view(np.uint32)), v
    q = np.fromfile(qf, dtype=np.float32).reshape(m, d)
    gt, _ = oracle.brute_force(vec, q, k)
    # the serial graph on the same rows (same generator seed): quality reference
    if M <= 16:
        _tool(*common, "--index", idx_s)
        gs = oracle.Graph(str(idx_s))
        for ef in (10, 60):
            rb = oracle.recall(gb.query_k(q, k, ef)[0], gt)
            rs = oracle.recall(gs.query_k(q, k, ef)[0], gt)
            print(f"recall@{k} ef={ef}: batched {rb:.4f}, serial {rs:.4f}")
            assert rb > rs - 0.04, (ef, rb, rs)
    else:
        assert oracle.recall(gb.query_k(q, k, 60)[0], gt) > 0.9
    from expann_amd import AntitopoEngine
    eng = AntitopoEngine(M, efc, 1, po, False, dim=d)
    eng.load_index(idx_b)
    eng.set_ef_search(40)
    ids, dists = eng.query_many(q, k)
    oids, od, _ = gb.query_k(q, k, 40)
    assert np.array_equal(ids, oids) and np.array_equal(dists.view(np.uint32), od.view(np.uint32))
    eng.close()


@pytest.mark.gpu
def test_c4_scale_properties(tmp_path):
    """N = 100 000 rows in the reference's configuration (M = 60, M0 = 120, ef_construction = 480),
    built by the batched GPU builder, searched with ef_search = 60, both bottom-layer variants:
    size-independent properties of every result list (ascending distances, no duplicate ids, ids in
    range, distcomps > 0, recall in the range this data allows) -- the oracle's walk at this size is
    covered on a sample by bench.py --workload c4."""
    n, d, m, k = 100_000, 128, 2000, 10
    rf = tmp_path / "c4.results"
    lines = _tool("--n", n, "--m", m, "--d", d, "--k", k, "--M", 60, "--ef_construction", 480, "--data", "sift",
                  "--batched", 1024, "--index", tmp_path / "c4.index", "--results", rf, "--ef", "60")
    assert lines[0]["builder"] == "batched gpu" and lines[0]["n"] == n
    q = [x for x in lines if x["phase"] == "query"]
    assert len(q) == 2 and all(0.6 < x["recall"] <= 1.0 and x["distcomps_per_query"] > 1000 for x in q)
    raw = np.fromfile(rf, dtype=np.uint8)
    per = m * k * 12 + m * 4
    for c in range(2):
        ids = raw[c * per:c * per + m * k * 8].view(np.uint64).reshape(m, k)
        dd = raw[c * per + m * k * 8:c * per + m * k * 12].view(np.float32).reshape(m, k)
        dc = raw[c * per + m * k * 12:(c + 1) * per].view(np.uint32)
        assert ids.max() < n and (dc > 0).all()
        assert (np.diff(dd, axis=1) >= 0).all() or c == 1   # (uint8 walk: order of the uint8 distances, fp32 re-score)
        srt = np.sort(ids, axis=1)
        assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate ids (basic_bench.h:98-104)"


@pytest.mark.gpu
def test_ortho_count_above_one_takes_the_serial_builder(tmp_path, oracle):
    """ortho_count > 1 (several "ortho" entry points per layer, src/antitopo_engine.h:336-379,396-413): the
    batched GPU builder implements the reference sweep's ortho_count = 1 only, so store_many_vectors_batched
    hands such an engine to the serial restatement -- the index file is byte-identical to the serial
    engine's, and the GPU walk of it equals the oracle's."""
    from expann_amd import AntitopoEngine
    from graph_helpers import check_against_oracle
    rng = np.random.RandomState(5)
    base = _sift_like(rng, 1200, 128)
    q = _sift_like(rng, 48, 128)
    files = []
    engs = {}
    for batched in (False, True):
        eng = AntitopoEngine(16, 80, 3, 0, False, dim=128)
        if batched:
            eng.store_many_vectors_batched(base, False, 256)
        else:
            eng.store_many_vectors(base, False)
        eng.build()
        path = str(tmp_path / f"ortho3_{int(batched)}.index")
        eng.save_index(path)
        files.append(open(path, "rb").read())
        engs[batched] = eng
    assert files[0] == files[1]
    engc = AntitopoEngine(16, 80, 3, 0, True, dim=128)
    engc.load_index(str(tmp_path / "ortho3_1.index"))
    check_against_oracle(oracle, {False: engs[True], True: engc}, str(tmp_path / "ortho3_1.index"), q, 10, efs=(10, 40))
