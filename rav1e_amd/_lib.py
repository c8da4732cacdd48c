"""ctypes loader for the product library.  Fails loudly: no fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librav1e_hip.so")


class R1Plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_int32), ("alloc_height", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32), ("xorigin", C.c_int32),
                ("yorigin", C.c_int32), ("bytes_per_px", C.c_int32), ("bit_depth", C.c_int32)]


class R1QuantParams(C.Structure):
    _fields_ = [("qindex", C.c_uint8), ("bit_depth", C.c_uint8), ("is_intra", C.c_uint8),
                ("dc_delta_q", C.c_int8), ("ac_delta_q", C.c_int8), ("reserved", C.c_uint8 * 3)]


class R1CdefParams(C.Structure):
    _fields_ = [("y_strengths", C.c_uint8 * 8), ("uv_strengths", C.c_uint8 * 8),
                ("damping", C.c_uint8), ("bit_depth", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class R1CdefSearchParams(C.Structure):
    _fields_ = [("y_strengths", C.c_uint8 * 8), ("uv_strengths", C.c_uint8 * 8),
                ("damping", C.c_int32), ("bit_depth", C.c_int32), ("n_idx", C.c_int32),
                ("planes", C.c_int32), ("xdec", C.c_int32), ("ydec", C.c_int32),
                ("crop_w", C.c_int32), ("crop_h", C.c_int32), ("area_sb_w", C.c_int32),
                ("area_sb_h", C.c_int32), ("dist_scale", C.c_uint32 * 3)]


class R1MeStats(C.Structure):
    _fields_ = [("row", C.c_int16), ("col", C.c_int16), ("normalized_sad", C.c_uint32)]


class R1MeParams(C.Structure):
    _fields_ = [("w_in_b", C.c_int32), ("h_in_b", C.c_int32), ("stats_cols", C.c_int32),
                ("stats_rows", C.c_int32), ("bit_depth", C.c_int32), ("allow_hp", C.c_int32),
                ("allow_full_search", C.c_int32), ("me_range_scale", C.c_int32),
                ("lambda_", C.c_uint32 * 3), ("launch_mode", C.c_int32)]


class R1MeJob(C.Structure):
    _fields_ = [("org", R1Plane * 3), ("ref", R1Plane * 3), ("stats", C.c_void_p),
                ("prev", C.c_void_p), ("tile_x", C.c_int32), ("tile_y", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32)]


# every symbol include/rav1e_amd.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _pd = C.c_void_p, C.c_int, C.c_size_t, C.c_ssize_t
_PP = C.POINTER(R1Plane)
SYMBOLS = {
    "r1_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "r1_ctx_destroy": (None, [_vp]),
    "r1_last_error": (C.c_char_p, []),
    "r1_abi_version": (_i, []),
    "r1_dist_batch": (_i, [_vp, _i, _PP, _PP, _i, _i, _vp, _i, _vp, _vp]),
    "r1_plane_pad": (_i, [_vp, _PP, _i, _i, _i, _i, _vp]),
    "r1_plane_downsample": (_i, [_vp, _PP, _PP, _i, _i, _i, _i, _vp]),
    "r1_dist_scaled_batch": (_i, [_vp, _i, _PP, _PP, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "r1_fwd_txfm_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "r1_inv_txfm_add_batch": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "r1_quantize_batch": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(R1QuantParams), _i, _vp, _vp,
                               _vp, _vp]),
    "r1_quantize_rdo_batch": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(R1QuantParams), _i, _vp, _vp,
                                   _vp, _vp, _vp, _vp]),
    "r1_dequantize_batch": (_i, [_vp, _vp, _i, _i, C.POINTER(R1QuantParams), _i, _vp, _vp]),
    "r1_intra_edges_batch": (_i, [_vp, _PP, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "r1_predict_intra_batch": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "r1_cfl_ac_batch": (_i, [_vp, _PP, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "r1_cdef_find_dir_batch": (_i, [_vp, _PP, _vp, _i, _vp, _vp, _vp]),
    "r1_cdef_filter_block_batch": (_i, [_vp, _PP, _PP, _i, _i, _vp, _i, _vp]),
    "r1_cdef_filter_frame_plane": (_i, [_vp, _PP, _PP, _PP, _i, _i, _i, _i, _i, _vp, _i, _i, _i,
                                        _vp, _i, C.POINTER(R1CdefParams), _vp]),
    "r1_cdef_analyze_blocks": (C.c_longlong, [_i, _i]),
    "r1_cdef_analyze_frame": (_i, [_vp, _PP, _i, _i, _i, _i, _vp, _vp, _vp]),
    "r1_cdef_filter_frame_plane_dirs": (_i, [_vp, _vp, _vp, _PP, _PP, _i, _i, _i, _i, _i, _vp, _i, _i, _i,
                                             _vp, _i, C.POINTER(R1CdefParams), _vp]),
    "r1_cdef_strength_search_scratch_bytes": (C.c_longlong, [_i, _i]),
    "r1_cdef_strength_search": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "r1_cdef_lrf_trial_scratch_bytes": (C.c_longlong, [_i, _i, _i, _i, _i, _i, _i]),
    "r1_cdef_lrf_trial_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp]),
    "r1_cdef_apply_area": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "r1_estimate_intra_costs": (_i, [_vp, _PP, _vp, _vp]),
    "r1_estimate_inter_costs": (_i, [_vp, _PP, _PP, _vp, _vp, _vp]),
    "r1_importance_block_difference": (_i, [_vp, _PP, _PP, _vp, _vp]),
    "r1_mc_put_batch": (_i, [_vp, _PP, _i, _i, _vp, _i, _vp, _vp]),
    "r1_mc_prep_batch": (_i, [_vp, _PP, _i, _i, _vp, _i, _vp, _vp]),
    "r1_mc_avg_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "r1_mc_batch_mfma": (_i, [_vp, _i, _PP, _i, _i, _vp, _i, _vp, _vp]),
    "r1_comm_library": (C.c_char_p, []),
    "r1_me_status": (_i, [_vp, _i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "r1_comm_unique_id": (_i, [_vp]),
    "r1_comm_create": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "r1_comm_destroy": (None, [_vp]),
    "r1_comm_rank": (_i, [_vp]),
    "r1_comm_world": (_i, [_vp]),
    "r1_comm_allgather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "r1_comm_exchange_halos": (_i, [_vp, _PP, _vp, _i, _vp]),
    "r1_comm_allgather_tiles": (_i, [_vp, _PP, _vp, _vp]),
    "r1_ipc_export": (_i, [_vp, _vp, _sz, _vp]),
    "r1_ipc_peer_access": (_i, [_vp]),
    "r1_ipc_open": (_i, [_vp, _vp, C.POINTER(_vp)]),
    "r1_ipc_close": (_i, [_vp, _vp]),
    "r1_push_rects": (_i, [_vp, _PP, _vp, _i, _vp, _i, _vp]),
    "r1_comm_open_peer_planes": (_i, [_vp, _vp, _PP, _vp]),
    "r1_comm_plane_pool_open": (_i, [_vp, _vp, _vp, _i, _vp]),
    "r1_comm_plane_pool_close": (_i, [_vp, _vp, _i, _vp]),
    "r1_comm_close_peer_planes": (_i, [_vp, _vp, _vp]),
    "r1_comm_barrier": (_i, [_vp, _vp]),
    "r1_comm_push_tile": (_i, [_vp, _vp, _PP, _vp, _vp, _vp]),
    "r1_comm_push_halos": (_i, [_vp, _vp, _PP, _vp, _vp, _i, _vp]),
    "r1_comm_push_frame": (_i, [_vp, _vp, _PP, _vp, _vp, _i, _vp, _vp]),
    "r1_rdo_cand_batch": (_i, [_vp, _PP, _PP, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "r1_rdo_full_cand_batch": (_i, [_vp, _PP, _PP, _i, _i, _i, _vp, _i, C.POINTER(R1QuantParams),
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "r1_estimate_tile_motion_batch": (_i, [_vp, C.POINTER(R1MeJob), _i, C.POINTER(R1MeParams), _vp]),
    "r1_estimate_motion_batch": (_i, [_vp, C.POINTER(R1MeJob), C.POINTER(R1MeParams), _vp, _i, _i, _i,
                                      _i, _i, _vp, _vp]),
    "r1_deblock_plane": (_i, [_vp, _vp, _PP, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "r1_deblock_sse_plane": (_i, [_vp, _PP, _PP, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "r1_deblock_frame": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "r1_deblock_sse_frame": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "r1_deblock_pick_levels": (_i, [_vp, _vp, _i, _vp]),
    "r1_intra_satd_batch": (_i, [_vp, _PP, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "r1_prescreen_select_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "r1_update_block_importances_scratch_bytes": (C.c_longlong, [_i, _i]),
    "r1_update_block_importances": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, C.c_longlong,
                                         _vp]),
    "r1_rdo_pixel_cand_batch": (_i, [_vp, _PP, _PP, _i, _i, _i, _vp, _i, C.POINTER(R1QuantParams), _i,
                                     _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "r1_rdo_pred_cand_batch": (_i, [_vp, _PP, _vp, _i, _i, _i, _vp, _i, C.POINTER(R1QuantParams), _i,
                                    _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "r1_rdo_txsearch_batch": (_i, [_vp, _PP, _PP, _vp, _i, _i, _i, _vp, _i, C.c_uint32, C.POINTER(R1QuantParams),
                                   _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "r1_tx_type_mask": (C.c_uint32, [_i, _i, _i, _i]),
    "r1_lrf_sgrproj_plane": (_i, [_vp, _PP, _PP, _PP, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "r1_sgrproj_solve_batch": (_i, [_vp, _PP, _PP, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "r1_lrf_search_batch": (_i, [_vp, _PP, _PP, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, C.c_uint32, _vp, _vp, _vp, _vp]),
    "r1_activity_scales": (_i, [_vp, _PP, _vp, _vp, _vp]),
    "r1_cfl_alpha_search_batch": (_i, [_vp, _PP, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rav1e_sad_hip": (C.c_uint32, [_vp, _pd, _vp, _pd, _i, _i]),
    "rav1e_satd_hip": (C.c_uint32, [_vp, _pd, _vp, _pd, _i, _i]),
    "rav1e_sad_hbd_hip": (C.c_uint32, [_vp, _pd, _vp, _pd, _i, _i]),
    "rav1e_satd_hbd_hip": (C.c_uint32, [_vp, _pd, _vp, _pd, _i, _i, C.c_uint32]),
    "rav1e_put_8tap_hip": (None, [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i]),
    "rav1e_put_8tap_hbd_hip": (None, [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i, _i]),
    "rav1e_inv_txfm_add_hip": (_i, [_vp, _pd, _vp, _i, _i, _i]),
    "rav1e_inv_txfm_add_hbd_hip": (_i, [_vp, _pd, _vp, _i, _i, _i, _i]),
    "rav1e_cdef_dir_hip": (_i, [_vp, _pd, _vp]),
    "rav1e_cdef_dir_hbd_hip": (_i, [_vp, _pd, _vp, _i]),
    "rav1e_cdef_filter_hip": (None, [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i]),
    "rav1e_cdef_filter_hbd_hip": (None, [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i, _i]),
    "rav1e_ipred_hip": (_i, [_vp, _pd, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "rav1e_fwd_txfm_hip": (_i, [_vp, _vp, _sz, _i, _i, _i, _i]),
}

_lib = None


def load():
    """Load librav1e_hip.so and bind every declared symbol.  Loading works
    without a GPU (symbols only resolve); calling into it needs a device."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError(
            "rav1e_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % SO)
    L = C.CDLL(SO)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(L, name)  # AttributeError if the export is missing
        f.restype, f.argtypes = res, args
    _lib = L
    return L
