"""Python module surface of the reference (upstream src/pyrunner.cpp:41-91) over the C ABI:
`AntitopoEngine(M, ef_construction, ortho_count, prune_overflow, use_compression)` with
store_vector / store_many_vectors(array2d, take_norms) / build / query_k / query_k_numpy /
set_ef_search / name / param_list, plus `query_many` (batched) and index save/load.

The reference compiles one module per dimension (expann_py_64/128/256/832/960,
CMakeLists.txt:102-153) and zero-pads every row to that DIM (src/pyrunner.cpp:20-27); here the
dimension is a constructor argument (default: the smallest supported multiple of 64 that holds
the first rows) and rows are zero-padded the same way.  take_norms = L2-normalise each row before
storing (angular data = normalise + L2, src/pyrunner.cpp:78-79); the normalisation itself is
Eigen's in the reference (summation order unpinned) and numpy float32 here.
"""
import ctypes as C

import numpy as np

from . import _lib

_SUPPORTED_DIMS = (64, 128, 256, 512, 768, 832, 960)


class AntitopoEngine:
    def __init__(self, M, ef_construction, ortho_count, prune_overflow, use_compression, dim=None,
                 device=0):
        self._L = _lib.load()
        self._args = (int(M), int(ef_construction), int(ortho_count), int(prune_overflow),
                      bool(use_compression))
        self.device = int(device)
        self.dim = None
        self._h = None
        if dim is not None:
            self._open(int(dim))

    def _open(self, dim):
        padded = next((d for d in _SUPPORTED_DIMS if d >= dim), None)
        if padded is None:
            raise ValueError(f"dimension {dim} exceeds the built graph kernels {_SUPPORTED_DIMS}")
        h = C.c_void_p()
        M, efc, oc, po, uc = self._args
        rc = self._L.expann_antitopo_create(padded, self.device, M, efc, oc, po, int(uc), C.byref(h))
        if rc != _lib.OK:
            raise _lib.ExpannError(rc, self._L.expann_antitopo_last_error(None).decode())
        self._h, self.dim = h, padded

    def _check(self, rc):
        if rc != _lib.OK:
            raise _lib.ExpannError(rc, self._L.expann_antitopo_last_error(self._h).decode())

    def _pad(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 1:
            a = a[None, :]
        if a.ndim != 2:
            raise RuntimeError("Input should be a 2D NumPy array")  # src/pyrunner.cpp:64-66
        if self._h is None:
            self._open(a.shape[1])
        if a.shape[1] > self.dim:
            raise ValueError("row longer than the engine dimension")
        if a.shape[1] < self.dim:  # convert_raw_to_eigen_padded, src/pyrunner.cpp:20-27
            p = np.zeros((a.shape[0], self.dim), dtype=np.float32)
            p[:, :a.shape[1]] = a
            a = p
        return a

    # ---- the reference's methods ------------------------------------------------------
    def name(self):
        return "GPU Anti-Topo Engine+ (MI355X)"

    def param_list(self):
        M, efc, oc, po, uc = self._args
        return {"M": str(M), "M0": str(2 * M), "ef_construction": str(efc), "ortho_count": str(oc),
                "prune_overflow": str(po), "use_compression": str(int(uc)),
                "num_distcomps": str(self._L.expann_antitopo_num_distcomps(self._h) if self._h else 0)}

    def store_vector(self, v):
        self.store_many_vectors(np.asarray(v, dtype=np.float32).reshape(1, -1), False)

    def store_many_vectors(self, array2d, take_norms):
        a = self._pad(array2d)
        if take_norms:
            a = a / np.sqrt(np.einsum("ij,ij->i", a, a, dtype=np.float32))[:, None]
            a = np.ascontiguousarray(a, dtype=np.float32)
        self._check(self._L.expann_antitopo_store(self._h, a.ctypes.data, a.shape[0]))

    def store_many_vectors_batched(self, array2d, take_norms=False, n_serial=2048):
        """store_many_vectors through the batched GPU builder (csrc/graph_build.hpp): the first
        n_serial rows of an empty engine are inserted serially, the rest in batches on the GPU."""
        a = self._pad(array2d)
        if take_norms:
            a = a / np.sqrt(np.einsum("ij,ij->i", a, a, dtype=np.float32))[:, None]
            a = np.ascontiguousarray(a, dtype=np.float32)
        self._check(self._L.expann_antitopo_store_batched(self._h, a.ctypes.data, a.shape[0], int(n_serial)))

    def build(self):
        if self._h is None:
            raise _lib.ExpannError(_lib.ERR_INVALID_ARG, "build() on an empty index")
        self._check(self._L.expann_antitopo_build(self._h))

    def query_k(self, v, k):
        ids, _ = self.query_many(np.asarray(v, dtype=np.float32).reshape(1, -1), k)
        row = ids[0]
        return [int(x) for x in row[row != np.uint64(2 ** 64 - 1)]]

    def query_k_numpy(self, array1d, k):
        return self.query_k(array1d, k)

    def set_ef_search(self, ef_search):
        self._check(self._L.expann_antitopo_set_ef_search(self._h, int(ef_search)))

    # ---- extensions ------------------------------------------------------------------
    def query_many(self, queries, k):
        q = self._pad(queries)
        ids = np.empty((q.shape[0], k), dtype=np.uint64)
        dists = np.empty((q.shape[0], k), dtype=np.float32)
        self._check(self._L.expann_antitopo_query(self._h, q.ctypes.data, q.shape[0], k,
                                                  ids.ctypes.data, dists.ctypes.data))
        return ids, dists

    def save_index(self, path):
        self._check(self._L.expann_antitopo_save(self._h, str(path).encode()))

    def load_index(self, path):
        self._check(self._L.expann_antitopo_load(self._h, str(path).encode()))

    def size(self):
        return self._L.expann_antitopo_size(self._h) if self._h else 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.expann_antitopo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
