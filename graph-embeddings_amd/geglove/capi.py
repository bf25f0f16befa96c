"""ctypes declarations for libgeglove.so -- one entry per symbol of include/geglove.h.

No torch types cross this boundary: plain pointers and sizes only.  Loading the library
does not touch the GPU; every computing entry point fails with GE_ERR_HIP when no gfx950
device is present (there is no CPU fallback in the product).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libgeglove.so")

GE_OK, GE_ERR_ARG, GE_ERR_OOM, GE_ERR_HIP, GE_ERR_STATE, GE_ERR_OVERFLOW = 0, -1, -2, -3, -4, -5
GE_COST_GLOVE, GE_COST_PGLOVE = 0, 1
GE_OPT_ADAGRAD, GE_OPT_ADAM, GE_OPT_AMSGRAD = 0, 1, 2
GE_NORM_NONE, GE_NORM_UNITY, GE_NORM_COUNTS = 0, 1, 2
GE_MODE_HOGWILD, GE_MODE_DETERMINISTIC = 0, 1
GE_SHUFFLE_JAVA, GE_SHUFFLE_DEVICE, GE_SHUFFLE_NONE = 0, 1, 2
GE_HOT_AUTO, GE_HOT_NONE, GE_HOT_ALL = 0, 1, 2
GE_DTYPE_F32, GE_DTYPE_BF16 = 0, 1
GE_LAYOUT_FIXED_CUTS, GE_LAYOUT_PLAIN_LONG_ROWS, GE_LAYOUT_SEPARATE_TABLES, GE_LAYOUT_PACKED_RECORDS, GE_LAYOUT_FIRST_PLACEMENT = 1, 2, 4, 8, 16
(GE_STATE_FOCUS, GE_STATE_CONTEXT, GE_STATE_FBIAS, GE_STATE_CBIAS, GE_STATE_GSQ_FOCUS,
 GE_STATE_GSQ_CONTEXT, GE_STATE_GSQ_FBIAS, GE_STATE_GSQ_CBIAS, GE_STATE_M2_FOCUS, GE_STATE_M2_CONTEXT,
 GE_STATE_M2_FBIAS, GE_STATE_M2_CBIAS) = range(12)
STATE_NAMES = ("focus", "context", "fbias", "cbias", "gsq_focus", "gsq_context", "gsq_fbias", "gsq_cbias")
M2_NAMES = ("m2_focus", "m2_context", "m2_fbias", "m2_cbias")
ALL_STATE_NAMES = STATE_NAMES + M2_NAMES

# every symbol include/geglove.h declares (tests check the library exports all of them)
SYMBOLS = (
    "ge_glove_cfg_default", "ge_glove_create", "ge_glove_epoch", "ge_glove_extract_f32",
    "ge_glove_extract_f64", "ge_glove_get_state", "ge_glove_set_state", "ge_glove_device_ptr",
    "ge_glove_epoch_order", "ge_glove_get_perm", "ge_glove_rng_state", "ge_glove_last_kernel_ms", "ge_glove_get_info", "ge_glove_destroy",
    "ge_bca_build", "ge_coo_get", "ge_coo_destroy", "ge_exchange_turn_bf16", "ge_glove_context_layout",
    "ge_local_group_create", "ge_local_group_destroy", "ge_local_group_abort", "ge_rccl_unique_id", "ge_rccl_selftest", "ge_sync_cfg_size", "ge_sync_create", "ge_sync_begin", "ge_sync_finish", "ge_sync_turn", "ge_sync_sync",
    "ge_sync_epoch", "ge_sync_hub_rows", "ge_sync_hub_exchange", "ge_sync_hub_exchange_live", "ge_sync_live_rows", "ge_sync_hub_plan", "ge_sync_replicate", "ge_sync_allreduce_f64", "ge_sync_destroy",
    "ge_sim_cfg_default", "ge_sim_cfg_size", "ge_sim_pattern_supported", "ge_similarity_pairs", "ge_sim_pairs_get", "ge_sim_pairs_destroy", "ge_copy_bandwidth", "ge_last_error", "ge_version", "ge_glove_cfg_size", "ge_bca_cfg_size", "ge_device_count",
)


class GloveCfg(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("dim", C.c_int32), ("nnz", C.c_int64),
                ("cost", C.c_int32), ("opt", C.c_int32), ("learning_rate", C.c_float),
                ("xmax", C.c_double), ("seed", C.c_int64), ("threads", C.c_int32),
                ("mode", C.c_int32), ("shuffle", C.c_int32), ("device", C.c_int32),
                ("stream", C.c_void_p), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("hot_columns", C.c_int32), ("workers", C.c_int32), ("emb_dtype", C.c_int32),
                ("hot_theta", C.c_float), ("stale_budget", C.c_float), ("flush_every", C.c_int32),
                ("blocks_per_cu", C.c_int32), ("layout_flags", C.c_int32)]


class GloveInfo(C.Structure):
    _fields_ = [("group_width", C.c_int32), ("vector_width", C.c_int32), ("chunks_per_lane", C.c_int32),
                ("blocks", C.c_int32), ("groups_in_flight", C.c_int32), ("hot_columns", C.c_int32),
                ("hot_nonzeros", C.c_int64), ("hot_threshold", C.c_int64), ("chunks", C.c_int64), ("hub_chunks", C.c_int64),
                ("long_rows", C.c_int64), ("shared_chunks", C.c_int64), ("flush_min", C.c_int32), ("row_stride", C.c_int32),
                ("runs", C.c_int64), ("schedule_bytes", C.c_int64), ("placements", C.c_int32), ("placement_best_ms", C.c_float),
                ("placement_worst_ms", C.c_float), ("reserved_", C.c_int32)]


class Csr(C.Structure):
    _fields_ = [("num_vertices", C.c_int32), ("ptr", C.POINTER(C.c_int64)),
                ("idx", C.POINTER(C.c_int32)), ("weight", C.POINTER(C.c_float))]


class BcaCfg(C.Structure):
    _fields_ = [("alpha", C.c_double), ("epsilon", C.c_double), ("directed", C.c_int32),
                ("normalize", C.c_int32), ("device", C.c_int32), ("row_begin", C.c_int32),
                ("row_end", C.c_int32), ("table_slots", C.c_int64), ("pool_entries", C.c_int64)]


class SimCfg(C.Structure):
    _fields_ = [("method", C.c_int32), ("threshold", C.c_double), ("ngram", C.c_int32), ("smooth", C.c_double),
                ("distance", C.c_double), ("time", C.c_int32), ("pattern", C.c_char_p), ("upper_triangle", C.c_int32),
                ("device", C.c_int32), ("job_begin", C.c_int32), ("job_end", C.c_int32)]


class Strings(C.Structure):
    _fields_ = [("count", C.c_int32), ("offset", C.POINTER(C.c_int64)), ("units", C.POINTER(C.c_uint16))]


class ContextLayout(C.Structure):
    _fields_ = [("table", C.c_void_p), ("dtype", C.c_int32), ("hub_rows", C.c_void_p), ("hub_index", C.c_void_p),
                ("n_hub", C.c_int32), ("vocab_size", C.c_int32), ("dim", C.c_int32), ("row_stride", C.c_int32), ("accum", C.c_void_p),
                ("accum_stride", C.c_int32), ("bias", C.c_void_p), ("bias_stride", C.c_int32), ("accum_bias", C.c_void_p),
                ("accum_bias_stride", C.c_int32)]


TRANSPORT_START = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p))
TRANSPORT_WAIT = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)
TRANSPORT_BCAST = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("start", TRANSPORT_START), ("wait", TRANSPORT_WAIT), ("broadcast", TRANSPORT_BCAST)]


class SyncCfg(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("wire", C.c_int32), ("accum_every", C.c_int32),
                ("transport", C.POINTER(Transport)), ("rccl_id", C.c_void_p), ("local_group", C.c_void_p)]


class GeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("geglove error %d: %s" % (status, message))
        self.status = status


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and its
    libraries ask for it by file name, so they load it even when /opt/rocm's copy is already in the process -- and
    the second runtime then sees no GPU ("No HIP GPUs are available").  The other order is fine: libgeglove.so asks
    for the SONAME and takes whichever copy is loaded.  So when torch is installed (the multi-GPU path and some
    tests use it in the same process) its copy is mapped first, without importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Loads libgeglove.so; raises (loudly) when the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libgeglove.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C graph-embeddings_amd/csrc`" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    i32p, i64p, f32p, f64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double))
    vp = C.c_void_p
    L.ge_glove_cfg_default.argtypes = [C.POINTER(GloveCfg)]; L.ge_glove_cfg_default.restype = None
    L.ge_glove_create.argtypes = [C.POINTER(GloveCfg), i32p, i32p, f32p, C.POINTER(vp)]
    L.ge_glove_epoch.argtypes = [vp, C.c_int32, f64p]
    L.ge_glove_extract_f32.argtypes = [vp, f32p]
    L.ge_glove_extract_f64.argtypes = [vp, f64p]
    L.ge_glove_get_state.argtypes = [vp, C.c_int32, f32p, C.c_int64]
    L.ge_glove_set_state.argtypes = [vp, C.c_int32, f32p, C.c_int64]
    L.ge_glove_device_ptr.argtypes = [vp, C.c_int32, C.POINTER(vp), i64p]
    L.ge_glove_epoch_order.argtypes = [vp, C.c_int32, i32p, C.c_int64]
    L.ge_glove_get_perm.argtypes = [vp, i32p, C.c_int64]
    L.ge_glove_rng_state.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ge_glove_last_kernel_ms.argtypes = [vp, f32p, i32p]
    L.ge_glove_get_info.argtypes = [vp, C.POINTER(GloveInfo)]
    L.ge_glove_destroy.argtypes = [vp]; L.ge_glove_destroy.restype = None
    L.ge_bca_build.argtypes = [C.POINTER(Csr), C.POINTER(Csr), C.POINTER(BcaCfg), C.POINTER(vp)]
    L.ge_coo_get.argtypes = [vp, i64p, C.POINTER(i32p), C.POINTER(i32p), C.POINTER(f32p), C.POINTER(i64p), f64p]
    L.ge_coo_destroy.argtypes = [vp]; L.ge_coo_destroy.restype = None
    L.ge_sim_cfg_default.argtypes = [C.POINTER(SimCfg)]; L.ge_sim_cfg_default.restype = None
    L.ge_sim_pattern_supported.argtypes = [C.c_char_p]; L.ge_sim_pattern_supported.restype = C.c_int32
    L.ge_similarity_pairs.argtypes = [C.POINTER(Strings), i32p, i32p, C.c_int32, i32p, i32p, C.c_int32, C.POINTER(SimCfg), C.POINTER(vp)]
    L.ge_sim_pairs_get.argtypes = [vp, i64p, C.POINTER(i32p), C.POINTER(i32p), C.POINTER(f32p)]
    L.ge_sim_pairs_destroy.argtypes = [vp]; L.ge_sim_pairs_destroy.restype = None
    L.ge_glove_context_layout.argtypes = [vp, C.POINTER(ContextLayout)]
    L.ge_exchange_turn_bf16.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, C.c_int32, C.c_uint32, vp]
    L.ge_rccl_unique_id.argtypes = [vp]
    L.ge_local_group_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.ge_local_group_destroy.argtypes = [vp]; L.ge_local_group_destroy.restype = None
    L.ge_local_group_abort.argtypes = [vp]; L.ge_local_group_abort.restype = None
    L.ge_rccl_selftest.argtypes = [C.c_int32]
    L.ge_sync_cfg_size.argtypes = []; L.ge_sync_cfg_size.restype = C.c_int32
    L.ge_sync_create.argtypes = [vp, C.POINTER(SyncCfg), C.POINTER(vp)]
    L.ge_sync_begin.argtypes = [vp, C.c_int32]
    for nm in ("ge_sync_finish", "ge_sync_turn", "ge_sync_sync"):
        getattr(L, nm).argtypes = [vp]
    L.ge_sync_epoch.argtypes = [vp, C.c_int32, C.c_int32, f64p]
    L.ge_sync_hub_rows.argtypes = [vp, i32p, C.c_int32, i32p]
    L.ge_sync_hub_exchange.argtypes = [vp]
    L.ge_sync_hub_exchange_live.argtypes = [vp]
    L.ge_sync_live_rows.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
    L.ge_sync_hub_plan.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ge_sync_replicate.argtypes = [vp, C.c_int32]
    L.ge_sync_allreduce_f64.argtypes = [vp, f64p, C.c_int32, C.c_int32]
    L.ge_sync_destroy.argtypes = [vp]; L.ge_sync_destroy.restype = None
    if L.ge_sync_cfg_size() != C.sizeof(SyncCfg):
        raise ImportError("libgeglove.so was built from another revision of include/geglove.h (ge_sync_cfg is %d bytes there, %d here)"
                          % (L.ge_sync_cfg_size(), C.sizeof(SyncCfg)))
    L.ge_copy_bandwidth.argtypes = [C.c_int32, C.c_int64, C.c_int32, f64p]
    L.ge_last_error.argtypes = []; L.ge_last_error.restype = C.c_char_p
    L.ge_version.argtypes = []; L.ge_version.restype = C.c_char_p
    L.ge_device_count.argtypes = []; L.ge_device_count.restype = C.c_int32
    L.ge_glove_cfg_size.argtypes = []; L.ge_glove_cfg_size.restype = C.c_int32
    L.ge_sim_cfg_size.argtypes = []; L.ge_sim_cfg_size.restype = C.c_int32
    L.ge_bca_cfg_size.argtypes = []; L.ge_bca_cfg_size.restype = C.c_int32
    if L.ge_bca_cfg_size() != C.sizeof(BcaCfg):
        raise ImportError("libgeglove.so was built from another revision of include/geglove.h (ge_bca_cfg is %d bytes there, %d here): "
                          "rebuild with `make -C graph-embeddings_amd/csrc`" % (L.ge_bca_cfg_size(), C.sizeof(BcaCfg)))
    if L.ge_sim_cfg_size() != C.sizeof(SimCfg):
        raise ImportError("libgeglove.so was built from another revision of include/geglove.h (ge_sim_cfg is %d bytes there, %d here): "
                          "rebuild with `make -C graph-embeddings_amd/csrc`" % (L.ge_sim_cfg_size(), C.sizeof(SimCfg)))
    if L.ge_glove_cfg_size() != C.sizeof(GloveCfg):
        raise ImportError("libgeglove.so was built from another revision of include/geglove.h (ge_glove_cfg is %d bytes there, "
                          "%d here): rebuild with `make -C graph-embeddings_amd/csrc`" % (L.ge_glove_cfg_size(), C.sizeof(GloveCfg)))
    for name in SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int:      # default restype -> ge_status
            f.restype = C.c_int32
    _lib = L
    return L


def check(status):
    if status != GE_OK:
        raise GeError(status, lib().ge_last_error().decode(errors="replace"))
