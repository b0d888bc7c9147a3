"""Helpers of the GPU graph tests: build an antitopo index from given rows through the C ABI
(expann_antitopo_*: host build, include/expann/antitopo_index.h), keep it as an index file in the
reference's layout, and compare the GPU traversal of that file with the oracle's walk."""
import numpy as np


def build_engines(base, tmp_path, M, ef_construction, prune_overflow=0, ortho_count=1):
    """({use_compression: AntitopoEngine}, index path): the fp32 engine builds and saves, the uint8
    one loads the same file."""
    from expann_amd import AntitopoEngine
    d = base.shape[1]
    idx = str(tmp_path / f"g_M{M}_po{prune_overflow}.index")
    eng = AntitopoEngine(M, ef_construction, ortho_count, prune_overflow, False, dim=d)
    eng.store_many_vectors(base, False)
    eng.build()
    eng.save_index(idx)
    engc = AntitopoEngine(M, ef_construction, ortho_count, prune_overflow, True, dim=d)
    engc.load_index(idx)
    return {False: eng, True: engc}, idx


def check_against_oracle(oracle, engs, idx, q, k, efs, compressions=(False, True)):
    """ids, distance bits and the RECORD_STATS distance-evaluation total of the GPU traversal ==
    the oracle's restatement of _query_k on the same index file.  Returns the oracle Graph."""
    g = oracle.Graph(idx)
    for comp in compressions:
        eng = engs[comp]
        for ef in efs:
            before = int(eng.param_list()["num_distcomps"])
            eng.set_ef_search(ef)
            ids, dists = eng.query_many(q, k)
            evals = int(eng.param_list()["num_distcomps"]) - before
            oids, od, odc = g.query_k(q, k, ef, comp)
            assert np.array_equal(ids, oids), (comp, ef)
            assert np.array_equal(dists.view(np.uint32), od.view(np.uint32)), (comp, ef)
            assert evals == int(odc.sum()), (comp, ef)
    return g


def read_index_degrees(path):
    """Per-vertex bottom-layer degree and the header fields of an index file in the reference's
    layout (src/antitopo_engine.h:932-991)."""
    import struct
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, raw, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]
    sv, M, M0, efm = take("QQQQ")
    if take("B"):
        take("Q")
    efc, oc = take("QQ")
    take("ff")
    po = take("Q")
    take("BB")
    max_layer = take("Q")
    n = take("Q")
    for _ in range(n):
        ln = take("Q")
        pos += 4 * ln
    nv = take("Q")
    deg0 = np.zeros(nv, dtype=np.int64)
    for v in range(nv):
        nl = take("Q")
        for layer in range(nl):
            ne = take("Q")
            if layer == 0:
                deg0[v] = ne
            pos += 12 * ne
    assert pos == len(raw)
    return dict(M=M, M0=M0, ef_construction=efc, prune_overflow=po, max_layer=max_layer, n=n,
                starting_vertex=sv), deg0


def read_index_edges(path):
    """(header, layers): layers[v][l] = (ids uint64[], dists float32[]) of an index file in the
    reference's layout (src/antitopo_engine.h:932-991)."""
    import struct
    raw = open(path, "rb").read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, raw, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]
    sv, M, M0, efm = take("QQQQ")
    if take("B"):
        take("Q")
    efc, oc = take("QQ")
    take("ff")
    po = take("Q")
    take("BB")
    max_layer = take("Q")
    n = take("Q")
    for _ in range(n):
        ln = take("Q")
        pos += 4 * ln
    nv = take("Q")
    rec = np.dtype([("d", "<f4"), ("id", "<u8")])
    layers = []
    for v in range(nv):
        nl = take("Q")
        per = []
        for layer in range(nl):
            ne = take("Q")
            a = np.frombuffer(raw, dtype=rec, count=ne, offset=pos)
            pos += 12 * ne
            per.append((a["id"].copy(), a["d"].copy()))
        layers.append(per)
    assert pos == len(raw)
    return dict(M=M, M0=M0, ef_construction=efc, prune_overflow=po, max_layer=max_layer, n=n,
                starting_vertex=sv), layers
