"""ctypes binding of oracle/liboracle.so (and oracle/_ref/libtopk_ref.so when built).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under expann_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

METRIC_L2_F32 = 0
METRIC_IP_F32 = 1
METRIC_L2_I8 = 2
METRIC_L2_I8_REFCOMPAT = 3
METRIC_IP_I8 = 4
METRIC_L2_U8 = 5
METRIC_L2_I16_REFCOMPAT = 6

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def build(arch_flags=None, out_dir=None):
    """Compile liboracle.so (and _ref/ when /root/reference exists).  Returns the .so path."""
    if arch_flags is None and out_dir is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
        return os.path.join(_HERE, "liboracle.so")
    out_dir = out_dir or _HERE
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liboracle_native.so")
    cmd = ["gcc", "-O3", "-fPIC", "-std=c11", "-fno-fast-math", "-ffp-contract=off", "-shared",
           "-o", out, os.path.join(_HERE, "expann_oracle.c"),
           os.path.join(_HERE, "expann_oracle_graph.c"), "-lm", "-lpthread"]
    cmd[2:2] = list(arch_flags or ["-march=x86-64-v3"])
    subprocess.check_call(cmd)
    return out


def _sig(lib):
    lib.oracle_l2_f32.restype = C.c_float
    lib.oracle_l2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.oracle_dot_f32.restype = C.c_float
    lib.oracle_dot_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for name in ("oracle_l2_i8_refcompat", "oracle_l2_i8", "oracle_l2_i16_refcompat",
                 "oracle_ip_i8", "oracle_l2_u8_compressed"):
        f = getattr(lib, name)
        f.restype = C.c_int32
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.oracle_quantize_simple_u8.restype = None
    lib.oracle_quantize_simple_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.oracle_quantize_ranged_q8.restype = None
    lib.oracle_quantize_ranged_q8.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                              _f32p, _f32p]
    lib.oracle_topk_create.restype = C.c_void_p
    lib.oracle_topk_create.argtypes = [C.c_size_t]
    lib.oracle_topk_destroy.restype = None
    lib.oracle_topk_destroy.argtypes = [C.c_void_p]
    lib.oracle_topk_consider.restype = C.c_int
    lib.oracle_topk_consider.argtypes = [C.c_void_p, C.c_float, C.c_uint64]
    lib.oracle_topk_discard_until_size.restype = None
    lib.oracle_topk_discard_until_size.argtypes = [C.c_void_p, C.c_size_t]
    lib.oracle_topk_size.restype = C.c_size_t
    lib.oracle_topk_size.argtypes = [C.c_void_p]
    lib.oracle_topk_worst.restype = C.c_uint64
    lib.oracle_topk_worst.argtypes = [C.c_void_p]
    lib.oracle_topk_worst_val.restype = C.c_float
    lib.oracle_topk_worst_val.argtypes = [C.c_void_p]
    lib.oracle_topk_at_capacity.restype = C.c_int
    lib.oracle_topk_at_capacity.argtypes = [C.c_void_p]
    lib.oracle_topk_to_combined.restype = C.c_size_t
    lib.oracle_topk_to_combined.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_brute_force_query_k.restype = C.c_size_t
    lib.oracle_brute_force_query_k.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                               C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_brute_force_batch.restype = None
    lib.oracle_brute_force_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                             C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p]
    lib.oracle_filter_by_score.restype = C.c_size_t
    lib.oracle_filter_by_score.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_size_t, C.c_float, C.c_void_p,
                                           C.c_void_p]
    lib.oracle_graph_load.restype = C.c_void_p
    lib.oracle_graph_load.argtypes = [C.c_char_p]
    lib.oracle_graph_destroy.restype = None
    lib.oracle_graph_destroy.argtypes = [C.c_void_p]
    lib.oracle_graph_size.restype = C.c_size_t
    lib.oracle_graph_size.argtypes = [C.c_void_p]
    lib.oracle_graph_dim.restype = C.c_size_t
    lib.oracle_graph_dim.argtypes = [C.c_void_p]
    lib.oracle_graph_vectors.restype = C.c_void_p
    lib.oracle_graph_vectors.argtypes = [C.c_void_p]
    lib.oracle_graph_query_k.restype = C.c_size_t
    lib.oracle_graph_query_k.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_heap_trace.restype = C.c_size_t
    lib.oracle_heap_trace.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t] + \
        [C.c_void_p] * 8
    lib.oracle_recall.restype = C.c_double
    lib.oracle_recall.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    return lib


_lib = None


def lib(path=None):
    global _lib
    if path is not None:
        return _sig(C.CDLL(path))
    if _lib is None:
        p = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(p):
            build()
        _lib = _sig(C.CDLL(p))
    return _lib


def ref_topk_lib():
    """The REFERENCE's topk_t<float> (oracle/_ref/libtopk_ref.so) or None when not built."""
    p = os.path.join(_HERE, "_ref", "libtopk_ref.so")
    if not os.path.exists(p):
        return None
    r = C.CDLL(p)
    r.ref_topk_run.restype = C.c_size_t
    r.ref_topk_run.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_long] + \
        [C.c_void_p] * 7
    return r


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_BASE_DT = {METRIC_L2_F32: np.float32, METRIC_IP_F32: np.float32, METRIC_L2_I8: np.int8,
            METRIC_L2_I8_REFCOMPAT: np.int8, METRIC_IP_I8: np.int8, METRIC_L2_U8: np.uint8,
            METRIC_L2_I16_REFCOMPAT: np.int16}
_Q_DT = dict(_BASE_DT)
_Q_DT[METRIC_L2_U8] = np.float32


def brute_force(base, queries, k, metric=METRIC_L2_F32, n_threads=1, _lib_override=None):
    """(ids[m,k] uint64, dists[m,k] float32); rows shorter than k padded with 2^64-1 / inf."""
    L = _lib_override or lib()
    base = np.ascontiguousarray(base, dtype=_BASE_DT[metric])
    queries = np.ascontiguousarray(queries, dtype=_Q_DT[metric])
    if queries.ndim == 1:
        queries = queries[None, :]
    n, d = base.shape
    m = queries.shape[0]
    ids = np.empty((m, k), dtype=np.uint64)
    dists = np.empty((m, k), dtype=np.float32)
    L.oracle_brute_force_batch(_ptr(base), n, d, _ptr(queries), m, k, metric, n_threads,
                               _ptr(ids), _ptr(dists))
    return ids, dists


def l2_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.float32(lib().oracle_l2_f32(_ptr(a), _ptr(b), a.size))


def dot_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.float32(lib().oracle_dot_f32(_ptr(a), _ptr(b), a.size))


def int_kernel(name, a, b):
    return int(getattr(lib(), name)(_ptr(a), _ptr(b), a.size))


def filter_by_score(base, query, ids, cutoff, metric=METRIC_L2_F32):
    base = np.ascontiguousarray(base, dtype=_BASE_DT[metric])
    query = np.ascontiguousarray(query, dtype=_Q_DT[metric])
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    kept = np.empty(ids.size, dtype=np.uint64)
    kd = np.empty(ids.size, dtype=np.float32)
    n = lib().oracle_filter_by_score(_ptr(base), base.shape[1], _ptr(query), metric, _ptr(ids),
                                     ids.size, cutoff, _ptr(kept), _ptr(kd))
    return kept[:n], kd[:n]


def recall(ans, expected):
    ans = np.ascontiguousarray(ans, dtype=np.uint64)
    expected = np.ascontiguousarray(expected, dtype=np.uint64)
    m, k = ans.shape
    return lib().oracle_recall(_ptr(ans), _ptr(expected), m, k)


class TopK:
    """oracle_topk_* object (restatement of src/topk_t.h)."""

    def __init__(self, k):
        self._h = lib().oracle_topk_create(k)
        self.k = k

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_topk_destroy(self._h)
            self._h = None

    def consider(self, d, v):
        return bool(lib().oracle_topk_consider(self._h, float(d), int(v)))

    def discard_until_size(self, goal):
        lib().oracle_topk_discard_until_size(self._h, goal)

    def size(self):
        return lib().oracle_topk_size(self._h)

    def worst(self):
        return lib().oracle_topk_worst(self._h)

    def worst_val(self):
        return lib().oracle_topk_worst_val(self._h)

    def at_capacity(self):
        return bool(lib().oracle_topk_at_capacity(self._h))

    def to_combined(self):
        n = self.size()
        ids = np.empty(n, dtype=np.uint64)
        d = np.empty(n, dtype=np.float32)
        lib().oracle_topk_to_combined(self._h, _ptr(ids), _ptr(d))
        return ids, d


def ref_topk_run(k, dists, ids, discard_goal=-1):
    """Run the REFERENCE topk_t<float> over a consider() sequence.  Returns a dict of traces."""
    r = ref_topk_lib()
    if r is None:
        raise RuntimeError("oracle/_ref/libtopk_ref.so not built")
    d = np.ascontiguousarray(dists, dtype=np.float32)
    v = np.ascontiguousarray(ids, dtype=np.uint64)
    n = d.size
    is_good = np.zeros(n, np.uint8)
    size_after = np.zeros(n, np.uint64)
    worst_after = np.zeros(n, np.uint64)
    worst_val_after = np.zeros(n, np.float32)
    at_cap = np.zeros(n, np.uint8)
    out_ids = np.zeros(k + 1, np.uint64)
    out_d = np.zeros(k + 1, np.float32)
    cnt = r.ref_topk_run(k, n, _ptr(d), _ptr(v), discard_goal, _ptr(is_good), _ptr(size_after),
                         _ptr(worst_after), _ptr(worst_val_after), _ptr(at_cap), _ptr(out_ids),
                         _ptr(out_d))
    assert cnt != 2 ** 64 - 1, "reference to_vector()/to_combined_vector() disagree"
    return dict(is_good=is_good, size_after=size_after, worst_after=worst_after,
                worst_val_after=worst_val_after, at_capacity_after=at_cap,
                out_ids=out_ids[:cnt], out_dists=out_d[:cnt])


def heap_trace(max_heap, init, ops, fn=None):
    """Run a queue trace through `fn` (default: the oracle's pq_* via oracle_heap_trace; any C
    function of the same signature, e.g. the std_heap hook of tests/native/).  init: [(d, id)],
    ops: [(1, d, id) | (0, 0, 0)].  Returns (states [(size, top_d_bits, top_id)], drain
    [(d_bits, id)]) in the format of tests/golden/heap_ref.json."""
    fn = fn or lib().oracle_heap_trace
    ini_d = np.array([x[0] for x in init], dtype=np.float32)
    ini_id = np.array([x[1] for x in init], dtype=np.uint64)
    op_k = np.array([x[0] for x in ops], dtype=np.int32)
    op_d = np.array([x[1] for x in ops], dtype=np.float32)
    op_id = np.array([x[2] for x in ops], dtype=np.uint64)
    n_ops = len(ops)
    size = np.zeros(n_ops + 1, dtype=np.uint64)
    top_d = np.zeros(n_ops + 1, dtype=np.float32)
    top_id = np.zeros(n_ops + 1, dtype=np.uint64)
    dr_d = np.zeros(len(init) + n_ops + 1, dtype=np.float32)
    dr_id = np.zeros(len(init) + n_ops + 1, dtype=np.uint64)
    nd = fn(int(max_heap), len(init), _ptr(ini_d), _ptr(ini_id), n_ops, _ptr(op_k), _ptr(op_d),
            _ptr(op_id), _ptr(size), _ptr(top_d), _ptr(top_id), _ptr(dr_d), _ptr(dr_id))
    tb = top_d.view(np.uint32)
    states = [(int(size[i]), int(tb[i]) if size[i] else 0, int(top_id[i]) if size[i] else 0)
              for i in range(n_ops + 1)]
    drain = [(int(dr_d.view(np.uint32)[i]), int(dr_id[i])) for i in range(nd)]
    return states, drain


class Graph:
    """oracle_graph_* (restatement of the query side of src/antitopo_engine.h) over an index
    file in the reference's binary layout."""

    def __init__(self, path):
        self._g = lib().oracle_graph_load(path.encode())
        if not self._g:
            raise IOError(f"cannot load index {path}")
        self.n = lib().oracle_graph_size(self._g)
        self.dim = lib().oracle_graph_dim(self._g)

    def vectors(self):
        p = C.cast(lib().oracle_graph_vectors(self._g), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(self.n, self.dim)).copy()

    def query_k(self, queries, k, ef_search, use_compression=False):
        """(ids[m,k] uint64 padded, dists[m,k], distcomps[m])"""
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        m = queries.shape[0]
        ids = np.full((m, k), 2 ** 64 - 1, dtype=np.uint64)
        dists = np.full((m, k), np.inf, dtype=np.float32)
        dc = np.zeros(m, dtype=np.uint64)
        for i in range(m):
            one = C.c_uint64()
            lib().oracle_graph_query_k(self._g, queries[i].ctypes.data, k, ef_search,
                                       int(use_compression), ids[i].ctypes.data,
                                       dists[i].ctypes.data, C.byref(one))
            dc[i] = one.value
        return ids, dists, dc

    def __del__(self):
        if getattr(self, "_g", None):
            lib().oracle_graph_destroy(self._g)
            self._g = None
