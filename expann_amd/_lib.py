"""ctypes binding of the C ABI in include/expann_hip.h (libexpann_hip.so, built in-tree by
__graft_entry__.build()).  There is no fallback: if the shared library is missing the import
of any compute entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXPANN_LIB") or os.path.join(_HERE, "libexpann_hip.so")  # (EXPANN_LIB: A/B runs of two builds)

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_BUILT, ERR_UNSUPPORTED, ERR_OVERFLOW = 1, 2, 3, 4, 5, 6
DTYPE_F32, DTYPE_U8, DTYPE_I8, DTYPE_I16 = 0, 1, 2, 3
METRIC_L2, METRIC_IP, METRIC_L2_I8_REFCOMPAT = 0, 1, 2

# every symbol include/expann_hip.h declares
ABI_SYMBOLS = [
    "expann_abi_version", "expann_device_count", "expann_create", "expann_destroy",
    "expann_last_error", "expann_add", "expann_build", "expann_set_base_device", "expann_size",
    "expann_search", "expann_search_device", "expann_sync", "expann_merge_topk_device", "expann_merge_topk_strided_device", "expann_score_ids",
    "expann_set_profiling", "expann_get_profile", "expann_set_option",
    "expann_quantize_simple_u8_device", "expann_quantize_ranged_q8_device",
    "expann_graph_create", "expann_graph_destroy", "expann_graph_last_error",
    "expann_graph_search", "expann_graph_last_kernel_ms",
    "expann_antitopo_create", "expann_antitopo_destroy", "expann_antitopo_last_error",
    "expann_antitopo_store", "expann_antitopo_build", "expann_antitopo_set_ef_search",
    "expann_antitopo_query", "expann_antitopo_save", "expann_antitopo_load",
    "expann_antitopo_size", "expann_antitopo_num_distcomps",
    "expann_graph_build_batched", "expann_antitopo_store_batched",
    "expann_sharded_create", "expann_sharded_unique_id", "expann_sharded_create_rank",
    "expann_sharded_destroy", "expann_sharded_last_error", "expann_sharded_add", "expann_sharded_build",
    "expann_sharded_set_shard_device", "expann_sharded_size", "expann_sharded_shards",
    "expann_sharded_exchange", "expann_sharded_set_exchange_fn", "expann_sharded_search", "expann_sharded_search_device",
    "expann_sharded_sync", "expann_sharded_set_option", "expann_sharded_set_profiling",
    "expann_sharded_get_profile", "expann_sharded_exchange_pattern", "expann_sharded_comm_ranks",
    "expann_sharded_last_enqueue_ms", "expann_sharded_set_alltoallv_fn", "expann_sharded_search_devices",
    "expann_sharded_slice", "expann_device_heap_trace",
]


# expann_exchange_fn(ctx, d_send, d_recv, bytes, rank, world, stream) -> 0 = ok
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
# expann_alltoallv_fn(ctx, d_send, send_off[], send_bytes[], d_recv, recv_off[], recv_bytes[], rank, world, stream)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p,
                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_void_p)


class Profile(C.Structure):
    _fields_ = [("scan_launches", C.c_uint64), ("scan_ms", C.c_double), ("scan_rows", C.c_uint64),
                ("scan_query_tiles", C.c_uint64), ("query_tile", C.c_uint32),
                ("levels", C.c_uint32), ("candidates", C.c_uint64), ("retries", C.c_uint64),
                ("scan_kernel", C.c_char * 64), ("deferred_searches", C.c_uint64)]


class ExpannError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"expann_hip error {code}: {msg}")
        self.code = code


_lib = None


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    PyTorch's ROCm wheels ship their OWN libamdhip64.so / libhsa-runtime64.so / librccl.so under
    torch/lib (file name `libamdhip64.so`, SONAME `libamdhip64.so.7`), while libexpann_hip.so asks
    for `libamdhip64.so.7` and finds /opt/rocm's.  The dynamic loader reuses an already loaded
    object only when the REQUESTED name equals its file name or SONAME:
      * torch first, library second: `libamdhip64.so.7` matches the SONAME of torch's copy -> one
        runtime (how every bench / test of round 1 happened to run);
      * library first, torch second: torch's `libamdhip64.so` matches neither name of /opt/rocm's
        copy -> a SECOND HIP + HSA runtime is mapped, and the second one to initialise finds the
        GPU already taken ("No HIP GPUs are available", the round-1 anomaly).
    So when a torch installation with a bundled runtime exists and has not been loaded yet, its
    runtime libraries are opened here first (by path, without importing torch): the library then
    binds to them by SONAME, and a later `import torch` finds the very files already mapped."""
    if "libamdhip64" in open("/proc/self/maps").read():
        return  # a runtime is mapped already (torch imported first, or a second call)
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if not spec or not spec.origin:
        return
    tlib = os.path.join(os.path.dirname(spec.origin), "lib")
    # (librccl.so is bundled the same way.  It is NOT opened here: pulled in ahead of torch's own load
    # order it brings torch's rocm_smi / roctx copies with it and the process then aborts in its exit
    # handlers ("double free or corruption", measured).  Library first + torch second therefore
    # leaves two independent RCCL copies -- the library's communicators in /opt/rocm's, torch's in its
    # own -- which works; torch first, as bench.py imports, gives one.)
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(tlib, name)
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def hip_runtimes_mapped(stem="libamdhip64"):
    """Distinct libamdhip64 (or `stem`) files mapped into this process (must be 1 once the library is loaded)."""
    return sorted({ln.split()[-1] for ln in open("/proc/self/maps") if stem in ln})


def load():
    """Load libexpann_hip.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc, gfx950).  There is no CPU fallback.")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    L.expann_abi_version.restype = C.c_int
    L.expann_device_count.restype = C.c_int
    L.expann_create.restype = C.c_int
    L.expann_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.expann_destroy.restype = None
    L.expann_destroy.argtypes = [vp]
    L.expann_last_error.restype = C.c_char_p
    L.expann_last_error.argtypes = [vp]
    L.expann_add.restype = C.c_int
    L.expann_add.argtypes = [vp, vp, sz]
    L.expann_build.restype = C.c_int
    L.expann_build.argtypes = [vp]
    L.expann_set_base_device.restype = C.c_int
    L.expann_set_base_device.argtypes = [vp, vp, sz, u64]
    L.expann_size.restype = sz
    L.expann_size.argtypes = [vp]
    L.expann_search.restype = C.c_int
    L.expann_search.argtypes = [vp, vp, sz, sz, vp, vp]
    L.expann_sync.restype = C.c_int
    L.expann_sync.argtypes = [vp]
    L.expann_search_device.restype = C.c_int
    L.expann_search_device.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.expann_merge_topk_device.restype = C.c_int
    L.expann_merge_topk_device.argtypes = [C.c_int, vp, vp, sz, sz, sz, vp, vp, vp]
    L.expann_merge_topk_strided_device.restype = C.c_int
    L.expann_merge_topk_strided_device.argtypes = [C.c_int, vp, vp, sz, sz, sz, sz, sz, vp, vp, vp]
    L.expann_score_ids.restype = C.c_int
    L.expann_score_ids.argtypes = [vp, vp, vp, sz, C.c_float, vp, vp, C.POINTER(sz)]
    L.expann_quantize_simple_u8_device.restype = C.c_int
    L.expann_quantize_simple_u8_device.argtypes = [C.c_int, vp, sz, vp, vp]
    L.expann_quantize_ranged_q8_device.restype = C.c_int
    L.expann_quantize_ranged_q8_device.argtypes = [C.c_int, vp, sz, vp, vp, vp]
    L.expann_graph_create.restype = C.c_int
    L.expann_graph_create.argtypes = [C.c_int, C.c_int, vp, sz, C.c_uint32, C.c_uint32, vp, vp,
                                      C.POINTER(vp)]
    L.expann_graph_destroy.restype = None
    L.expann_graph_destroy.argtypes = [vp]
    L.expann_graph_last_error.restype = C.c_char_p
    L.expann_graph_last_error.argtypes = [vp]
    L.expann_graph_search.restype = C.c_int
    L.expann_graph_search.argtypes = [vp, vp, sz, sz, sz, C.c_int, vp, vp, vp]
    L.expann_graph_last_kernel_ms.restype = C.c_double
    L.expann_graph_last_kernel_ms.argtypes = [vp]
    L.expann_antitopo_create.restype = C.c_int
    L.expann_antitopo_create.argtypes = [C.c_int, C.c_int, sz, sz, sz, sz, C.c_int, C.POINTER(vp)]
    L.expann_antitopo_destroy.restype = None
    L.expann_antitopo_destroy.argtypes = [vp]
    L.expann_antitopo_last_error.restype = C.c_char_p
    L.expann_antitopo_last_error.argtypes = [vp]
    L.expann_antitopo_store.restype = C.c_int
    L.expann_antitopo_store.argtypes = [vp, vp, sz]
    L.expann_antitopo_store_batched.restype = C.c_int
    L.expann_antitopo_store_batched.argtypes = [vp, vp, sz, sz]
    L.expann_antitopo_build.restype = C.c_int
    L.expann_antitopo_build.argtypes = [vp]
    L.expann_antitopo_set_ef_search.restype = C.c_int
    L.expann_antitopo_set_ef_search.argtypes = [vp, sz]
    L.expann_antitopo_query.restype = C.c_int
    L.expann_antitopo_query.argtypes = [vp, vp, sz, sz, vp, vp]
    L.expann_antitopo_save.restype = C.c_int
    L.expann_antitopo_save.argtypes = [vp, C.c_char_p]
    L.expann_antitopo_load.restype = C.c_int
    L.expann_antitopo_load.argtypes = [vp, C.c_char_p]
    L.expann_antitopo_size.restype = sz
    L.expann_antitopo_size.argtypes = [vp]
    L.expann_antitopo_num_distcomps.restype = u64
    L.expann_antitopo_num_distcomps.argtypes = [vp]
    L.expann_sharded_create.restype = C.c_int
    L.expann_sharded_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.expann_sharded_unique_id.restype = C.c_int
    L.expann_sharded_unique_id.argtypes = [vp]
    L.expann_sharded_create_rank.restype = C.c_int
    L.expann_sharded_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                             C.POINTER(vp)]
    L.expann_sharded_destroy.restype = None
    L.expann_sharded_destroy.argtypes = [vp]
    L.expann_sharded_last_error.restype = C.c_char_p
    L.expann_sharded_last_error.argtypes = [vp]
    L.expann_sharded_add.restype = C.c_int
    L.expann_sharded_add.argtypes = [vp, vp, sz]
    L.expann_sharded_build.restype = C.c_int
    L.expann_sharded_build.argtypes = [vp]
    L.expann_sharded_set_shard_device.restype = C.c_int
    L.expann_sharded_set_shard_device.argtypes = [vp, C.c_int, vp, sz, u64]
    L.expann_sharded_size.restype = sz
    L.expann_sharded_size.argtypes = [vp]
    L.expann_sharded_shards.restype = C.c_int
    L.expann_sharded_shards.argtypes = [vp]
    L.expann_sharded_exchange.restype = C.c_int
    L.expann_sharded_exchange.argtypes = [vp]
    L.expann_sharded_set_exchange_fn.restype = C.c_int
    L.expann_sharded_set_exchange_fn.argtypes = [vp, EXCHANGE_FN, vp]
    L.expann_sharded_set_alltoallv_fn.restype = C.c_int
    L.expann_sharded_set_alltoallv_fn.argtypes = [vp, ALLTOALLV_FN, vp]
    L.expann_sharded_exchange_pattern.restype = C.c_int
    L.expann_sharded_exchange_pattern.argtypes = [vp]
    L.expann_sharded_comm_ranks.restype = C.c_int
    L.expann_sharded_comm_ranks.argtypes = [vp]
    L.expann_sharded_last_enqueue_ms.restype = C.c_double
    L.expann_sharded_last_enqueue_ms.argtypes = [vp]
    L.expann_sharded_search_devices.restype = C.c_int
    L.expann_sharded_search_devices.argtypes = [vp, C.POINTER(vp), sz, sz, C.POINTER(vp), C.POINTER(vp)]
    L.expann_sharded_slice.restype = C.c_int
    L.expann_sharded_slice.argtypes = [vp, sz, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.expann_sharded_search.restype = C.c_int
    L.expann_sharded_search.argtypes = [vp, vp, sz, sz, vp, vp]
    L.expann_sharded_search_device.restype = C.c_int
    L.expann_sharded_search_device.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.expann_sharded_sync.restype = C.c_int
    L.expann_sharded_sync.argtypes = [vp]
    L.expann_sharded_set_option.restype = C.c_int
    L.expann_sharded_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.expann_sharded_set_profiling.restype = C.c_int
    L.expann_sharded_set_profiling.argtypes = [vp, C.c_int]
    L.expann_sharded_get_profile.restype = C.c_int
    L.expann_sharded_get_profile.argtypes = [vp, C.c_int, C.POINTER(Profile)]
    L.expann_set_profiling.restype = C.c_int
    L.expann_set_profiling.argtypes = [vp, C.c_int]
    L.expann_get_profile.restype = C.c_int
    L.expann_get_profile.argtypes = [vp, C.POINTER(Profile)]
    L.expann_set_option.restype = C.c_int
    L.expann_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    _lib = L
    return L


def check(handle, rc):
    if rc != OK:
        msg = load().expann_last_error(handle)
        raise ExpannError(rc, msg.decode() if msg else "")
