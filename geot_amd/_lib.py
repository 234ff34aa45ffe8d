"""ctypes binding of libgeot_hip.so (the C ABI declared in include/geot_hip.h).

This is the reference-side stub a maintainer would write to bind the library
(see INTEGRATION.md): plain pointers and sizes, no torch types cross the
boundary.  The library is the ONLY compute path of this package: if it is
missing or fails to load, every op raises -- there is no CPU or PyTorch
fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GEOT_DISTANCE selects the squared-distance arithmetic of every index-producing op (include/geot_hip.h
# geot_distance_mode): "exact" (default) | "fma" | "fma_xy"; one library per mode, chosen once at load time.
DISTANCE_MODES = {"exact": (0, "libgeot_hip.so"), "fma": (1, "libgeot_hip_fma.so"), "fma_xy": (2, "libgeot_hip_fma_xy.so")}
DISTANCE = os.environ.get("GEOT_DISTANCE", "exact")
if DISTANCE not in DISTANCE_MODES:
    raise ImportError("GEOT_DISTANCE must be one of %s, got %r" % (sorted(DISTANCE_MODES), DISTANCE))
LIB_PATH = os.path.join(_HERE, DISTANCE_MODES[DISTANCE][1])

_c_int, _c_float, _c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_P = ctypes.c_void_p  # device pointers are passed as raw addresses

# name -> argtypes (all return int = hipError_t), in include/geot_hip.h order.
PROTOTYPES = {
    "geot_furthest_point_sampling": [_c_int, _c_int, _c_int, _P, _P, _P, _c_int, _c_int, _c_void_p],
    "geot_furthestsampling_offset": [_c_int, _c_int, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_gather_points": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_gather_points_grad": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_ball_query": [_c_int, _c_int, _c_int, _c_float, _c_int, _P, _P, _P, _c_void_p],
    "geot_ballquery_offset": [_c_int, _c_int, _c_float, _c_int, _P, _P, _P, _P, _P, _c_void_p],
    "geot_group_points": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_group_points_grad": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_three_nn": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_three_interpolate": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_three_interpolate_grad": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_knnquery_heap": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_knn_sorted": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_grouping_cl": [_c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_grouping_cl_grad": [_c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_interpolation_cl": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_interpolation_cl_grad": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_subtraction_cl": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_subtraction_cl_grad": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_aggregation_cl": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_void_p],
    "geot_aggregation_cl_grad": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_sa_group_mlp_max": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_float, _c_int,
                              ctypes.POINTER(_c_int), _c_int, _P, _P, _c_void_p],
}
PROTOTYPES.update({
    "geot_gather_points_grad_ws": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_group_points_grad_ws": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_three_interpolate_grad_ws": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_void_p],
    "geot_three_interpolate_grad_out": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_void_p],
    "geot_fp_weights": [_c_int, _c_int, _P, _P, _c_void_p],
    "geot_three_interpolate_into": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_three_interpolate_grad_from": [_c_int, _c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, _P, _P, _P, _c_void_p],
    "geot_ball_query_ws": [_c_int, _c_int, _c_int, _c_float, _c_int, _P, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_knnquery_heap_ws": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_knn_sorted_nd": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_knn_sorted_ws": [_c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_three_nn_ws": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_graph_feature": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_graph_feature_grad": [_c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_class_anchors": [_c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_ntm_class_transition": [_c_int, ctypes.c_double, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_class_transition_grad": [_c_int, ctypes.c_double, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_sig_t_mean_grad_w": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_sig_t_mean": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_sig_t_mean_grad_raw": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_correct": [_c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_correct_grad": [_c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_correct_grad_ws": [_c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_threed_loss": [_c_int, _c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_threed_loss_grad": [_c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_threed_loss_grad_ws": [_c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _P, _P, _P, _P, _P, _P, _P,
                                     ctypes.c_longlong, _c_void_p],
    "geot_ntm_threed_loss_ord": [_c_int, _c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_threed_loss_fwd_graph": [_c_int, _c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _P, _P,
                                       ctypes.c_longlong, _c_void_p],
    "geot_ntm_threed_loss_grad_graph": [_c_int, _c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, ctypes.c_longlong,
                                        _P, _c_void_p],
    "geot_grid_subsampling": [_c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_pc_norm_stats": [_c_int, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_cloud_sample": [_c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _P, _P, _c_void_p],
    "geot_spatial_order": [_c_int, _c_int, _P, _P, _P, ctypes.c_longlong, _c_void_p],
    "geot_ntm_feature_loss": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _P, _P, _P, _P, _P, _c_void_p],
    "geot_ntm_feature_loss_grad": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _P, _P, _P, _P, _P,
                                   _c_void_p],
})
PROTOTYPES.update({
    "geot_softmax_grad": [ctypes.c_longlong, _c_int, _P, _P, _P, _c_void_p],
    "geot_qkv_split": [_c_int] * 4 + [_c_float] + [_P] * 2 + [_c_void_p],
    "geot_qkv_split_grad": [_c_int] * 4 + [_c_float] + [_P] * 4 + [_c_void_p],
    "geot_res_ln": [_c_int] * 3 + [_c_float] + [_P] * 10 + [_c_void_p],
    "geot_res_ln_grad": [_c_int] * 3 + [_P] * 12 + [_c_void_p],
    "geot_poly1_focal": [_c_int] * 3 + [_c_float] * 3 + [_P] * 5 + [_c_void_p],
    "geot_poly1_focal_grad": [_c_int] * 3 + [_c_float] * 3 + [_P] * 6 + [_c_void_p],
    "geot_bn_sums": [_c_int] * 3 + [_P] * 2 + [_c_void_p],
    "geot_bn_sums_shifted": [_c_int] * 3 + [_P] * 2 + [_c_void_p],
    "geot_bn_sums_shifted_cl": [_c_int, _c_int, _P, _P, _c_void_p],
    "geot_bn_finalize": [_c_int, _P, ctypes.c_double, _P, ctypes.c_double, ctypes.c_double] + [_P] * 9 + [_c_void_p],
    "geot_bn_bwd_coef": [_c_int, _P, _P, ctypes.c_double, _P] + [_P] * 4 + [_c_void_p],
    "geot_bn_stats": [_c_int] * 3 + [_P] * 2 + [_c_void_p],
    "geot_bn_apply": [_c_int] * 4 + [_P] * 4 + [_c_void_p],
    "geot_bn_bwd_reduce": [_c_int] * 4 + [_P] * 7 + [_c_void_p],
    "geot_bn_bwd_apply": [_c_int] * 4 + [_P] * 10 + [_c_void_p],
    "geot_fp_front": [_c_int] * 5 + [_P] * 7 + [_c_void_p],
    "geot_fp_front_cl": [_c_int] * 5 + [_P] * 8 + [_c_void_p],
    "geot_bn_stats_cl": [ctypes.c_longlong, _c_int, _P, _P, _c_void_p],
    "geot_bn_apply_cl": [ctypes.c_longlong, _c_int, _c_int] + [_P] * 4 + [_c_void_p],
    "geot_bn_bwd_reduce_cl": [ctypes.c_longlong, _c_int, _c_int] + [_P] * 7 + [_c_void_p],
    "geot_bn_bwd_apply_cl": [ctypes.c_longlong, _c_int, _c_int] + [_P] * 10 + [_c_void_p],
    "geot_bn_sums_cl": [_c_int, _c_int, _P, _P, _c_void_p],
    "geot_rix_build": [_c_int] * 4 + [_P] * 4 + [ctypes.c_longlong, _c_void_p],
    "geot_gather_rows_csr_cl": [_c_int] * 5 + [_P] * 4 + [_c_void_p],
    "geot_bn_sums_k_cl": [_c_int] * 3 + [_P] * 2 + [_c_void_p],
    "geot_fp_skip_wgrad_cl": [_c_int, _c_int] + [_P] * 6 + [_c_void_p],
    "geot_bn_bwd_reduce_skip_cl": [_c_int] * 5 + [_P] * 8 + [_c_void_p],
    "geot_gather_rows_csr_bn_cl": [_c_int] * 6 + [_P] * 11 + [_c_void_p],
    "geot_segment_max": [ctypes.c_longlong, _c_int, _P, _P, _P, _c_void_p],
    "geot_segment_sum": [ctypes.c_longlong, _c_int, _P, _P, _c_void_p],
    "geot_bn_pool": [_c_int] * 5 + [_P] * 6 + [_c_void_p],
    "geot_bn_pool_grad": [_c_int] * 4 + [_P] * 9 + [_c_void_p],
    "geot_rowdot_small": [_c_int, _c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_colsum": [_c_int, _c_int, _P, _P, _P, _c_void_p],
    "geot_segment_max_grad": [ctypes.c_longlong, _c_int, _P, _P, _P, _c_void_p],
    "geot_edgeconv_gn_max": [_c_int] * 6 + [_c_float, _c_float] + [_P] * 11 + [ctypes.c_longlong, _c_void_p],
    "geot_edgeconv_gn_max_grad": [_c_int] * 6 + [_c_float] + [_P] * 15 + [ctypes.c_longlong, _c_void_p],
    "geot_edgeconv_gn_max_grad_rix": [_c_int] * 6 + [_c_float] + [_P] * 15 + [ctypes.c_longlong, _c_void_p],
    "geot_edgeconv_rix_build": [_c_int] * 4 + [_P, _P, ctypes.c_longlong, _c_void_p],
})
PROTOTYPES["geot_rowsum_f64"] = [ctypes.c_longlong, _c_int, _P, _P, _c_void_p]
# entry points that do not follow the "(..., stream) -> hipError_t" shape
PLAIN = {
    "geot_sa_param_floats": ([_c_int, _c_int, ctypes.POINTER(_c_int)], _c_int),
    "geot_knn_grid_ws_bytes": ([_c_int, _c_int], ctypes.c_longlong),
    "geot_knnquery_heap_ws_bytes": ([_c_int, _c_int, _c_int, _c_int], ctypes.c_longlong),
    "geot_grad_ws_needs_zero": ([_c_int, _c_int, _c_int, ctypes.c_longlong, _c_int], _c_int),
    "geot_scatter_grad_ws_floats": ([_c_int, _c_int, _c_int, ctypes.c_longlong, _c_int, _c_int], ctypes.c_longlong),
    "geot_ntm_sig_t_mean_ws_floats": ([_c_int, _c_int], ctypes.c_longlong),
    "geot_ntm_threed_loss_ws_bytes": ([_c_int, _c_int, _c_int], ctypes.c_longlong),
    "geot_ntm_correct_ws_floats": ([_c_int, _c_int], ctypes.c_longlong),
    "geot_ntm_threed_graph_bytes": ([_c_int, _c_int, _c_int], ctypes.c_longlong),
    "geot_grid_subsampling_ws_bytes": ([_c_int], ctypes.c_longlong),
    "geot_pc_norm_ws_bytes": ([], ctypes.c_longlong),
    "geot_knn_grid_eligible": ([_c_int, _c_int, _c_int, _c_int], _c_int),
    "geot_ball_grid_eligible": ([_c_int, _c_int, _c_int, _c_float, _c_int], _c_int),
    "geot_edgeconv_eligible": ([_c_int] * 6, _c_int),
    "geot_bn_slices": ([_c_int] * 3, _c_int),
    "geot_fp_front_slices": ([_c_int] * 4, _c_int),
    "geot_cl_tiles": ([_c_int, ctypes.c_longlong, _c_int], _c_int),
    "geot_fp_front_cl_tiles": ([_c_int] * 4, _c_int),
    "geot_cl_stat_floats": ([_c_int, _c_int], ctypes.c_longlong),
    "geot_rix_ws_ints": ([_c_int, ctypes.c_longlong, _c_int, _c_int], ctypes.c_longlong),
    "geot_edgeconv_ws_bytes": ([_c_int] * 5, ctypes.c_longlong),
    "geot_edgeconv_rix_ints": ([_c_int] * 4, ctypes.c_longlong),
    "geot_poly1_focal_ws_doubles": ([_c_int] * 3, ctypes.c_longlong),
    "geot_res_ln_supported": ([_c_int], _c_int),
    "geot_res_ln_ws_floats": ([_c_int] * 2, ctypes.c_longlong),
    "geot_rowdot_small_slices": ([_c_int] * 2, _c_int),
    "geot_colsum_ws_floats": ([_c_int] * 2, ctypes.c_longlong),
}
ABI_VERSION = 7     # include/geot_hip.h GEOT_ABI_VERSION this binding was written against

_lib = None


class GeotLibraryError(RuntimeError):
    pass


def load():
    """Load libgeot_hip.so once.  Raises GeotLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GeotLibraryError(
            "geot_amd: %s is missing -- build it with `python -m geot_amd.build %s` "
            "(there is no CPU/PyTorch fallback)" % (LIB_PATH, DISTANCE))
    # Make sure the HIP runtime torch already uses is the one the library binds
    # to (same SONAME libamdhip64.so.7 => the loader reuses the loaded copy), so
    # torch's stream handles are valid in our launches.
    try:
        import torch  # noqa: F401
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(tl):
            ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    except ImportError:
        pass
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise GeotLibraryError("geot_amd: cannot load %s: %s" % (LIB_PATH, e))
    lib.geot_abi_version.restype = _c_int
    lib.geot_abi_version.argtypes = []
    if lib.geot_abi_version() != ABI_VERSION:
        raise GeotLibraryError("geot_amd: %s is ABI version %d, this package binds version %d -- rebuild it with "
                               "`python -m geot_amd.build`" % (LIB_PATH, lib.geot_abi_version(), ABI_VERSION))
    lib.geot_error_string.restype = ctypes.c_char_p
    lib.geot_error_string.argtypes = [_c_int]
    lib.geot_distance_mode.restype = _c_int
    lib.geot_distance_mode.argtypes = []
    if lib.geot_distance_mode() != DISTANCE_MODES[DISTANCE][0]:
        raise GeotLibraryError("geot_amd: %s was built with distance mode %d, GEOT_DISTANCE=%s needs %d" %
                               (LIB_PATH, lib.geot_distance_mode(), DISTANCE, DISTANCE_MODES[DISTANCE][0]))
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = _c_int
        fn.argtypes = argtypes
    for name, (argtypes, restype) in PLAIN.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(err, what):
    if err != 0:
        msg = load().geot_error_string(err)
        raise RuntimeError("geot_amd: %s failed: %s (hipError %d)" %
                           (what, msg.decode() if msg else "?", err))


def exported_symbols():
    """All C-ABI symbol names the Python side binds."""
    return ["geot_abi_version", "geot_distance_mode", "geot_error_string"] + list(PROTOTYPES) + list(PLAIN)
