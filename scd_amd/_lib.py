"""ctypes binding of libscd_hip.so (include/scd_hip.h).

The library is the product: if it is missing and cannot be built, importing fails
loudly - there is no CPU or torch fallback for any hot-path op.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("SCD_HIP_LIB") or os.path.join(_HERE, "lib", "libscd_hip.so")   # override: A/B runs of two builds

SCD_OK, SCD_EINVAL, SCD_EHIP, SCD_ERCCL, SCD_EINFEASIBLE = 0, -1, -2, -3, -4
SCD_F32, SCD_F16 = 0, 1
SIM_RAW, SIM_SOFTMAX = 0, 1


class ScdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libscd_hip: %s (status %d)" % (msg, code))
        self.code = code


class EncoderDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("mlp_dim", C.c_int),
                ("tokens", C.c_int), ("patch", C.c_int), ("image", C.c_int), ("vocab", C.c_int), ("out_dim", C.c_int),
                ("act", C.c_int), ("ln_eps", C.c_float)]


_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
# scd_exchange_fn (include/scd_hip.h): int (*)(void* ctx, double* buf, int64_t n_doubles, void* stream)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
# scd_gather_fn: int (*)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank, void* stream)


class LloydRestart(C.Structure):
    """scd_lloyd_restart (include/scd_hip.h): one restart's handle and buffers for scd_kmeans_lloyd_run_multi."""
    _fields_ = [("h", C.c_void_p), ("lab_ring", C.c_void_p), ("labels_prev", C.c_void_p), ("C_start", C.c_void_p), ("C_ring", C.c_void_p),
                ("sums", C.c_void_p), ("counts", C.c_void_p), ("stats_ring", C.c_void_p), ("best_labels", C.c_void_p), ("best_C", C.c_void_p),
                ("result_host", C.c_void_p), ("ws_e", C.c_void_p), ("ws_m", C.c_void_p)]


GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

# name -> (restype, argtypes); every symbol declared in include/scd_hip.h
SIGNATURES = {
    "scd_version": (_i, []),
    "scd_last_error": (C.c_char_p, []),
    "scd_create": (_i, [_i, C.POINTER(_vp)]),
    "scd_destroy": (_i, [_vp]),
    "scd_trace_mark": (_i, [_vp, _i, _vp]),
    "scd_l2norm_rows": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp]),
    "scd_sim_topk_ws_bytes": (_sz, [_i64, _i, _i64, _i]),
    "scd_sim_topk": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scd_sim_vocab_norm": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "scd_sim_topk_prenorm": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "scd_sim_argmax": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _f, _vp, _vp, _vp, _sz, _vp]),
    "scd_transpose_f16": (_i, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "scd_gather_rows_f16": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "scd_select_rows": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "scd_mean2_f16": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "scd_prompt_pool": (_i, [_vp, _vp, _i, _i, _i, _i64, _i64, _vp, _vp]),
    "scd_kmeans_prep_bytes": (_sz, [_i64, _i]),
    "scd_kmeans_prepare": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "scd_kmeans_estep_ws_bytes": (_sz, [_i64, _i, _i]),
    "scd_kmeans_estep": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "scd_kmeans_estep_hint": (_i, [_vp, _i]),
    "scd_kmeans_timing": (_i, [_vp, _i, _vp, _i, _vp]),
    "scd_kmeans_rowdist": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "scd_kmeans_dist": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "scd_kmeans_mstep_ws_bytes": (_sz, [_i64, _i, _i]),
    "scd_kmeans_mstep": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scd_f16_exact": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "scd_f16_exact_max": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "scd_kmeans_mstep_f16": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scd_kmeans_finalize": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _i64, _vp]),
    "scd_labels_changed": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "scd_kpp_searchsorted": (_i, [_vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "scd_kmeans_min_update": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "scd_kpp_draw_ws_bytes": (_sz, [_i64]),
    "scd_kpp_draw": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scd_kmeans_sumsq": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _vp, _vp]),
    "scd_kmeans_lloyd_step_delta": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz,
                                         _vp, _sz, _vp]),
    # h, X_u, prep, n_u, X16, n_cat, d, k, labels_lab, lab_ring, labels_prev, C_start, C_ring, sums, counts, sums_lab, counts_lab, sumsq4,
    # stats_ring, max_iter, tol, best_labels, best_C, result_host, ws_e, nb_e, ws_m, nb_m, stream
    "scd_kmeans_lloyd_run": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_double,
                                  _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    # ... the same, then xbuf, exchange (EXCHANGE_FN), exchange_ctx
    "scd_kmeans_lloyd_run_sharded": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                          C.c_double, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, EXCHANGE_FN, _vp]),
    "scd_kmeans_lloyd_step": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "scd_kmeans_min_update_multi": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i64, _vp]),
    "scd_kpp_draw_multi": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scd_kpp_seed_ws_bytes": (_sz, [_i64, _i, _i]),
    # h, X, X16, n, d, R, d2, ld, r_dev, T, C_buf, k, m0, picks_out, ws, nb, stream
    "scd_kpp_seed_lockstep": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i64, _vp, _i, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "scd_kpp_seed_sharded_ws_bytes": (_sz, [_i64, _i, _i]),
    "scd_kpp_seed_sharded_xbuf_bytes": (_sz, [_i, _i, _i]),
    # restarts, R, X_u, prep_u, n_u, X16_cat, n_cat, d, k, labels_lab, sums_lab, counts_lab, sumsq4, max_iter, tol, ws_e_bytes, ws_m_bytes, stream,
    # xbuf, exchange (EXCHANGE_FN or NULL), exchange_ctx, n_streams
    "scd_kmeans_lloyd_run_multi": (_i, [_vp, _i, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i, C.c_double, _sz, _sz, _vp, _vp, _vp, _vp, _i]),
    # h, X, X16, n, d, R, d2, ld, r_dev, T, C_buf, k, m0, picks_out, ws, nb, stream, xbuf, xbuf_bytes, gather (GATHER_FN), ctx, rank, world
    "scd_kpp_seed_lockstep_sharded": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i64, _vp, _i, _vp, _i, _i, _vp, _vp, _sz, _vp, _vp, _sz,
                                           GATHER_FN, _vp, _i, _i]),
    "scd_kpp_greedy_ws_bytes": (_sz, [_i64, _i, _i, _i]),
    # h, X, X16, n, d, R, L, k, first, u, C_buf, picks_out, ws, nb, stream
    "scd_kpp_greedy_lockstep": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    # h, X, prep, n, X16, d, k, lab_ring, labels_prev, C_start, C_ring, sums, counts, sumsq4, stats_ring, max_iter, tol, final_labels,
    # final_C, result_host, ws_e, nb_e, ws_m, nb_m, stream
    "scd_kmeans_lloyd_run_sk": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_double, _vp, _vp, _vp,
                                     _vp, _sz, _vp, _sz, _vp]),
    "scd_kpp_update_ws_bytes": (_sz, [_i64, _i]),
    # h, X16, n, d, R, c_new, d2, ld, first_call, ws, nb, stream
    "scd_kpp_update_filter": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _i64, _i, _vp, _sz, _vp]),
    "scd_sum_f32_multi": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp]),
    "scd_sum_f32": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "scd_vote_hist_ws_bytes": (_sz, [_i64, _i]),
    "scd_vote_hist": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "scd_vote_table": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _i, _i64, _i64, _i, _vp, _vp, _vp]),
    "scd_vote_table_topm": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp]),
    "scd_munkres": (_i, [_vp, _i, _i, _vp, C.POINTER(_i)]),
    "scd_munkres_sparse": (_i, [_i, _i64, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "scd_transport_solve": (_i, [_vp, _i64, _i, _i, _i, _vp, C.POINTER(_i64)]),
    "scd_transport_solve_batch": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _i]),
    "scd_comm_unique_id_bytes": (_sz, []),
    "scd_comm_unique_id": (_i, [_vp]),
    "scd_comm_init": (_i, [_vp, _i, _i, _vp]),
    "scd_comm_destroy": (_i, [_vp]),
    "scd_allreduce_centroids": (_i, [_vp, _vp, _i64, _vp]),
    "scd_allgather_text": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "scd_encoder_create": (_i, [_vp, C.POINTER(EncoderDesc), C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "scd_encoder_destroy": (_i, [_vp]),
    "scd_encoder_ws_bytes": (_sz, [_vp, _i]),
    "scd_encoder_timing": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double)]),
    "scd_vit_encode_image": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "scd_clip_encode_text": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "scd_clip_encode_text_len": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "scd_gemm_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
}

_lock = threading.Lock()
_lib = None


def lib_path():
    return _LIB_PATH


def load():
    """Load (building first if the .so is absent and hipcc is available)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            from . import build as _build
            _build.build(verbose=False)
        # torch first: it ships a HIP runtime of its own, and whichever libamdhip64 is loaded first is the process's only one (same SONAME).
        # Loaded before torch, this library pulled in /opt/rocm's runtime and torch's later initialisation left it without a device
        # (round 6: __graft_entry__.build() followed by smoke() in one process failed in hipGetDeviceCount).  Hosts without torch
        # (examples/c_abi_host.cpp, a foreign-language binding) load the library directly and are not affected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def check(status):
    if status != SCD_OK:
        raise ScdError(status, load().scd_last_error().decode("utf-8", "replace"))


_handles = {}


def handle(device=None):
    """One scd_handle per device (rank)."""
    import torch
    if not torch.cuda.is_available():
        raise ScdError(SCD_EHIP, "no HIP device visible: the scd_amd hot path has no CPU fallback")
    if device is None:
        device = torch.cuda.current_device()
    device = int(device)
    h = _handles.get(device)
    if h is None:
        out = _vp()
        check(load().scd_create(device, C.byref(out)))
        h = _handles[device] = out
    return h


_extra_handles = {}


def extra_handles(n, device=None):
    """n further scd_handles of the device (scd_kmeans_lloyd_run_multi: every restart needs a handle of its own - scratch, centre
    hand-over and statistics ring are per handle).  Created once and kept."""
    import torch
    if device is None:
        device = torch.cuda.current_device()
    device = int(device)
    hs = _extra_handles.setdefault(device, [])
    while len(hs) < n:
        out = _vp()
        check(load().scd_create(device, C.byref(out)))
        hs.append(out)
    return hs[:n]


def stream_ptr():
    import torch
    return _vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Raw device (or host) pointer of a tensor / numpy array; None -> NULL."""
    if t is None:
        return _vp(0)
    if hasattr(t, "data_ptr"):
        return _vp(t.data_ptr())
    return _vp(t.ctypes.data)
