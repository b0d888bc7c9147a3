"""Host-side mirror (Python) of the reference's engine interface for the brute-force path.

`GpuBruteForceEngine` has the five operations of the reference's CRTP `ann_engine`
(upstream src/ann_engine.h:16-29: name / param_list / store_vector / build / query_k) with
the meaning `brute_force_engine` gives them (src/brute_force_engine.h:9-46), plus the batch
and device-pointer extensions the GPU path needs.  All compute goes through the C ABI
(include/expann_hip.h); this module only moves pointers.  The C++ mirror a maintainer would
drop into the reference tree is include/expann/gpu_brute_force_engine.h.
"""
import ctypes as C

import numpy as np

from . import _lib

_NP_DTYPE = {_lib.DTYPE_F32: np.float32, _lib.DTYPE_U8: np.uint8, _lib.DTYPE_I8: np.int8,
             _lib.DTYPE_I16: np.int16}


class GpuBruteForceEngine:
    """Exact k-NN by a full scan on one MI355X (drop-in for brute_force_engine<float>)."""

    def __init__(self, dim, metric="l2", dtype="f32", device=0):
        self._L = _lib.load()
        self.dim = int(dim)
        self.metric = {"l2": _lib.METRIC_L2, "ip": _lib.METRIC_IP,
                       "l2_i8_refcompat": _lib.METRIC_L2_I8_REFCOMPAT}[metric]
        self.dtype = {"f32": _lib.DTYPE_F32, "u8": _lib.DTYPE_U8, "i8": _lib.DTYPE_I8,
                      "i16": _lib.DTYPE_I16}[dtype]
        self.device = int(device)
        self._metric_name, self._dtype_name = metric, dtype
        h = C.c_void_p()
        rc = self._L.expann_create(self.dim, self.dtype, self.metric, self.device, C.byref(h))
        _lib.check(None, rc)
        self._h = h

    # ---- the reference interface (src/ann_engine.h:16-29) ---------------------------
    def name(self):
        return "GPU Brute-Force Engine (MI355X)"

    def param_list(self):
        return {"device": str(self.device), "metric": self._metric_name, "dtype": self._dtype_name}

    def store_vector(self, v):
        """src/brute_force_engine.h:20-22: copies one row; ids are insertion order."""
        self.store_many_vectors(np.asarray(v).reshape(1, -1))

    def store_many_vectors(self, rows):
        rows = np.ascontiguousarray(rows, dtype=_NP_DTYPE[self.dtype])
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}]")
        _lib.check(self._h, self._L.expann_add(self._h, rows.ctypes.data, rows.shape[0]))

    def build(self):
        """src/brute_force_engine.h:24-26 (asserts a non-empty index): upload to HBM."""
        _lib.check(self._h, self._L.expann_build(self._h))

    def query_k(self, v, k):
        """src/brute_force_engine.h:28-46: ids of the min(k, n) nearest rows, ascending."""
        ids, _ = self.query_k_batch(np.asarray(v).reshape(1, -1), k)
        row = ids[0]
        return [int(x) for x in row[row != np.uint64(2 ** 64 - 1)]]

    # ---- extensions ---------------------------------------------------------------
    def query_k_batch(self, queries, k):
        """(ids[m,k] uint64, dists[m,k] float32); short rows padded with 2^64-1 / +inf."""
        qdt = np.float32 if self.dtype in (_lib.DTYPE_F32, _lib.DTYPE_U8) else _NP_DTYPE[self.dtype]
        queries = np.ascontiguousarray(queries, dtype=qdt)
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"queries must be [m, {self.dim}]")
        m = queries.shape[0]
        ids = np.empty((m, k), dtype=np.uint64)
        dists = np.empty((m, k), dtype=np.float32)
        _lib.check(self._h, self._L.expann_search(self._h, queries.ctypes.data, m, k,
                                                  ids.ctypes.data, dists.ctypes.data))
        return ids, dists

    def set_base_device(self, ptr, n, id_offset=0):
        _lib.check(self._h, self._L.expann_set_base_device(self._h, C.c_void_p(ptr), n, id_offset))

    def search_device(self, q_ptr, m, k, ids_ptr, dists_ptr, stream=0):
        _lib.check(self._h, self._L.expann_search_device(
            self._h, C.c_void_p(q_ptr), m, k, C.c_void_p(ids_ptr), C.c_void_p(dists_ptr),
            C.c_void_p(stream)))

    def sync(self):
        """expann_sync: wait for the searches enqueued under set_option("async_search", 1) and
        report on them (raises if one needs the synchronous retry)."""
        _lib.check(self._h, self._L.expann_sync(self._h))

    def score_ids(self, query, ids, cutoff=float("inf")):
        """quantized_scorer::filter_by_score (src/quantizer.h:20-59): (kept_ids, kept_scores)."""
        qdt = np.float32 if self.dtype in (_lib.DTYPE_F32, _lib.DTYPE_U8) else _NP_DTYPE[self.dtype]
        query = np.ascontiguousarray(query, dtype=qdt)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        kept = np.empty(ids.size, dtype=np.uint64)
        sc = np.empty(ids.size, dtype=np.float32)
        n = C.c_size_t()
        _lib.check(self._h, self._L.expann_score_ids(self._h, query.ctypes.data, ids.ctypes.data,
                                                     ids.size, cutoff, kept.ctypes.data,
                                                     sc.ctypes.data, C.byref(n)))
        return kept[:n.value], sc[:n.value]

    def set_option(self, name, value):
        _lib.check(self._h, self._L.expann_set_option(self._h, name.encode(), int(value)))

    def set_profiling(self, enable=True):
        _lib.check(self._h, self._L.expann_set_profiling(self._h, int(bool(enable))))

    def get_profile(self):
        p = _lib.Profile()
        _lib.check(self._h, self._L.expann_get_profile(self._h, C.byref(p)))
        return {f: (getattr(p, f).decode() if f == "scan_kernel" else getattr(p, f))
                for f, _ in _lib.Profile._fields_}

    def size(self):
        return self._L.expann_size(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.expann_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedBruteForceEngine:
    """Row-sharded exact k-NN over several GPUs behind ONE handle (expann_sharded_*): rows cut into
    contiguous ranges, one RCCL all-gather of the per-shard top-k, merge in the reference's (score, id)
    order -- results bit-identical to GpuBruteForceEngine.  Two forms:
      ShardedBruteForceEngine(dim, devices=[0, 1, ...])                  in-process, host buffers
      ShardedBruteForceEngine(dim, device=d, rank=r, world=G, unique_id=b)   one process per GPU"""

    def __init__(self, dim, metric="l2", dtype="f32", devices=None, device=0, rank=None, world=None,
                 unique_id=None):
        self._L = _lib.load()
        self.dim = int(dim)
        self.metric = {"l2": _lib.METRIC_L2, "ip": _lib.METRIC_IP,
                       "l2_i8_refcompat": _lib.METRIC_L2_I8_REFCOMPAT}[metric]
        self.dtype = {"f32": _lib.DTYPE_F32, "u8": _lib.DTYPE_U8, "i8": _lib.DTYPE_I8,
                      "i16": _lib.DTYPE_I16}[dtype]
        h = C.c_void_p()
        if rank is None:
            devs = list(devices if devices is not None else [device])
            arr = (C.c_int * len(devs))(*devs)
            rc = self._L.expann_sharded_create(self.dim, self.dtype, self.metric, arr, len(devs), C.byref(h))
            self.devices, self.rank, self.world = devs, None, len(devs)
        else:
            buf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
            rc = self._L.expann_sharded_create_rank(self.dim, self.dtype, self.metric, int(device), int(rank),
                                                    int(world), buf, C.byref(h))
            self.devices, self.rank, self.world = [int(device)], int(rank), int(world)
        if rc != _lib.OK:
            raise _lib.ExpannError(rc, self._L.expann_sharded_last_error(None).decode())
        self._h = h

    @staticmethod
    def unique_id():
        """128 bytes for the rank form (ncclGetUniqueId); rank 0 makes it, the launcher distributes it."""
        L = _lib.load()
        buf = C.create_string_buffer(128)
        rc = L.expann_sharded_unique_id(buf)
        if rc != _lib.OK:
            raise _lib.ExpannError(rc, L.expann_sharded_last_error(None).decode())
        return buf.raw

    def _check(self, rc):
        if rc != _lib.OK:
            raise _lib.ExpannError(rc, self._L.expann_sharded_last_error(self._h).decode())

    def name(self):
        return "GPU Brute-Force Engine, row-sharded (MI355X)"

    def param_list(self):
        return {"devices": ",".join(map(str, self.devices)), "shards": str(self.shards()),
                "exchange": {0: "none", 1: "rccl", 2: "device copies", 3: "caller"}[self.exchange()],
                "exchange_pattern": {0: "none", 1: "all-gather", 2: "all-to-all of query slices"}[self.exchange_pattern()]}

    def store_many_vectors(self, rows):
        rows = np.ascontiguousarray(rows, dtype=_NP_DTYPE[self.dtype])
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}]")
        self._check(self._L.expann_sharded_add(self._h, rows.ctypes.data, rows.shape[0]))

    def store_vector(self, v):
        self.store_many_vectors(np.asarray(v).reshape(1, -1))

    def build(self):
        self._check(self._L.expann_sharded_build(self._h))

    def set_shard_device(self, shard, ptr, n, id_offset):
        self._check(self._L.expann_sharded_set_shard_device(self._h, int(shard), C.c_void_p(ptr), n, id_offset))

    def query_k_batch(self, queries, k):
        qdt = np.float32 if self.dtype in (_lib.DTYPE_F32, _lib.DTYPE_U8) else _NP_DTYPE[self.dtype]
        queries = np.ascontiguousarray(queries, dtype=qdt)
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"queries must be [m, {self.dim}]")
        m = queries.shape[0]
        ids = np.empty((m, k), dtype=np.uint64)
        dists = np.empty((m, k), dtype=np.float32)
        self._check(self._L.expann_sharded_search(self._h, queries.ctypes.data, m, k, ids.ctypes.data,
                                                  dists.ctypes.data))
        return ids, dists

    def query_k(self, v, k):
        ids, _ = self.query_k_batch(np.asarray(v).reshape(1, -1), k)
        row = ids[0]
        return [int(x) for x in row[row != np.uint64(2 ** 64 - 1)]]

    def search_device(self, q_ptr, m, k, ids_ptr, dists_ptr, stream=0):
        self._check(self._L.expann_sharded_search_device(self._h, C.c_void_p(q_ptr), m, k, C.c_void_p(ids_ptr),
                                                         C.c_void_p(dists_ptr), C.c_void_p(stream)))

    def search_devices(self, q_ptrs, m, k, ids_ptrs, dists_ptrs):
        """expann_sharded_search_devices (in-process form, everything resident): q_ptrs[r] = the m queries
        on shard r's device; shard r leaves its merged query slice (self.slice(m, r)) at ids_ptrs[r] /
        dists_ptrs[r].  Deferred by default: sync() waits for every device and validates."""
        n = len(q_ptrs)
        arr = lambda ps: (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ps])
        self._check(self._L.expann_sharded_search_devices(self._h, arr(q_ptrs), m, k, arr(ids_ptrs), arr(dists_ptrs)))

    def slice(self, m, shard):
        """[lo, hi): the queries shard `shard` merges (exchange pattern 2, search_devices)."""
        lo, hi = C.c_size_t(), C.c_size_t()
        self._check(self._L.expann_sharded_slice(self._h, m, int(shard), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def sync(self):
        self._check(self._L.expann_sharded_sync(self._h))

    def exchange_pattern(self):
        """0 none, 1 all-gather of whole chunks, 2 all-to-all of query slices"""
        return self._L.expann_sharded_exchange_pattern(self._h)

    def comm_ranks(self):
        """ranks of the RCCL communicator as ncclCommCount reports them (0: no communicator)"""
        return self._L.expann_sharded_comm_ranks(self._h)

    def last_enqueue_ms(self):
        return self._L.expann_sharded_last_enqueue_ms(self._h)

    def set_alltoallv_fn(self, fn):
        """Rank form: fn(d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes, rank, world, stream)
        -> 0 (lists of `world` ints; entry [rank] is 0 bytes) moves the query slices of exchange pattern 2
        in place of RCCL's send / recv groups (expann_sharded_set_alltoallv_fn); None = RCCL again."""
        if fn is None:
            self._a2afn = _lib.ALLTOALLV_FN(0)
        else:
            def tramp(_ctx, d_send, so, sb, d_recv, ro, rb, rank, world, stream):
                try:
                    return int(fn(d_send or 0, [so[j] for j in range(world)], [sb[j] for j in range(world)],
                                  d_recv or 0, [ro[j] for j in range(world)], [rb[j] for j in range(world)],
                                  rank, world, stream or 0))
                except Exception:  # (an exception cannot cross the C frames above this one)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._a2afn = _lib.ALLTOALLV_FN(tramp)  # (kept alive with the engine)
        self._check(self._L.expann_sharded_set_alltoallv_fn(self._h, self._a2afn, None))

    def set_exchange_fn(self, fn):
        """Rank form: fn(d_send, d_recv, nbytes, rank, world, stream) -> 0 gathers every rank's chunk of
        device memory (expann_sharded_set_exchange_fn) in place of RCCL's all-gather; None = RCCL again."""
        if fn is None:
            self._xfn = _lib.EXCHANGE_FN(0)
        else:
            def tramp(_ctx, d_send, d_recv, nbytes, rank, world, stream):
                try:
                    return int(fn(d_send or 0, d_recv or 0, nbytes, rank, world, stream or 0))
                except Exception:  # (an exception cannot cross the C frames above this one)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._xfn = _lib.EXCHANGE_FN(tramp)  # (kept alive with the engine)
        self._check(self._L.expann_sharded_set_exchange_fn(self._h, self._xfn, None))

    def set_option(self, name, value):
        self._check(self._L.expann_sharded_set_option(self._h, name.encode(), int(value)))

    def set_profiling(self, enable=True):
        self._check(self._L.expann_sharded_set_profiling(self._h, int(bool(enable))))

    def get_profile(self, shard=0):
        p = _lib.Profile()
        self._check(self._L.expann_sharded_get_profile(self._h, int(shard), C.byref(p)))
        return {f: (getattr(p, f).decode() if f == "scan_kernel" else getattr(p, f))
                for f, _ in _lib.Profile._fields_}

    def size(self):
        return self._L.expann_sharded_size(self._h)

    def shards(self):
        return self._L.expann_sharded_shards(self._h)

    def exchange(self):
        return self._L.expann_sharded_exchange(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.expann_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_strided_device(device, in_ids_ptr, in_dists_ptr, ids_stride, dists_stride, n_lists, m,
                              k, out_ids_ptr, out_dists_ptr, stream=0):
    """expann_merge_topk_strided_device: list g at in_ids + g*ids_stride / in_dists + g*dists_stride
    (strides in elements)."""
    L = _lib.load()
    rc = L.expann_merge_topk_strided_device(device, C.c_void_p(in_ids_ptr), C.c_void_p(in_dists_ptr),
                                            ids_stride, dists_stride, n_lists, m, k,
                                            C.c_void_p(out_ids_ptr), C.c_void_p(out_dists_ptr),
                                            C.c_void_p(stream))
    _lib.check(None, rc)


def merge_topk_device(device, in_ids_ptr, in_dists_ptr, n_lists, m, k, out_ids_ptr,
                      out_dists_ptr, stream=0):
    L = _lib.load()
    rc = L.expann_merge_topk_device(device, C.c_void_p(in_ids_ptr), C.c_void_p(in_dists_ptr),
                                    n_lists, m, k, C.c_void_p(out_ids_ptr),
                                    C.c_void_p(out_dists_ptr), C.c_void_p(stream))
    _lib.check(None, rc)
