"""ctypes binding of include/crazyara_hip.h (the C ABI a cgo/JNI/C++ shim would bind)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcrazyara_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)

# name -> (restype, argtypes); kept in one table so tests can check that every symbol of the header is exported
SIGNATURES = {
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (C.c_char_p, []),
    "mi_device_count": (C.c_int, []),
    "mi_host_alloc": (C.c_void_p, [C.c_size_t]),
    "mi_host_free": (None, [C.c_void_p]),
    "mi_net_create": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_char_p]),
    "mi_net_destroy": (None, [C.c_void_p]),
    "mi_net_calibrate_int8": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_int]),
    "mi_net_has_int8_calibration": (C.c_int, [C.c_char_p]),
    "mi_onnx_to_cranet": (C.c_int, [C.c_char_p, C.c_char_p]),
    "mi_e4m3_from_float": (C.c_int, [C.c_float]),
    "mi_e5m2_from_float": (C.c_int, [C.c_float]),
    "mi_net_design": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "mi_net_model_name": (C.c_char_p, [C.c_void_p]),
    "mi_net_flops_per_position": (C.c_double, [C.c_void_p]),
    "mi_net_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_net_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_net_last_submit_zero_copy": (C.c_int, [C.c_void_p]),
    "mi_net_wait": (C.c_int, [C.c_void_p]),
    "mi_net_device_buffers": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5),
    "mi_net_keep_logits": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_net_block_dump": (C.c_void_p, [C.c_void_p, c_int_p]),
    "mi_net_forward_device": (C.c_int, [C.c_void_p]),
    "mi_net_sync": (C.c_int, [C.c_void_p]),
    "mi_net_stream": (C.c_void_p, [C.c_void_p]),
    "mi_net_time_forward": (C.c_int, [C.c_void_p, C.c_int, c_float_p]),
    "mi_net_op_count": (C.c_int, [C.c_void_p]),
    "mi_net_time_ops": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), c_float_p]),
    "mi_net_submit_boards": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_net_submit_boards_gathered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    # environment
    "mi_pos_create": (C.c_void_p, [C.c_char_p, C.c_int, C.c_char_p]),
    "mi_pos_clone": (C.c_void_p, [C.c_void_p]),
    "mi_pos_destroy": (None, [C.c_void_p]),
    "mi_pos_fen": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mi_pos_side_to_move": (C.c_int, [C.c_void_p]),
    "mi_pos_legal_moves": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "mi_pos_uci_to_move": (C.c_uint32, [C.c_void_p, C.c_char_p]),
    "mi_pos_move_to_uci": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_int]),
    "mi_pos_do_move": (C.c_int, [C.c_void_p, C.c_uint32]),
    "mi_pos_terminal": (C.c_int, [C.c_void_p]),
    "mi_pos_game_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mi_pos_number_repetitions": (C.c_int, [C.c_void_p]),
    "mi_pos_in_check": (C.c_int, [C.c_void_p]),
    "mi_pos_insufficient_material": (C.c_int, [C.c_void_p]),
    "mi_pos_plies_from_null": (C.c_int, [C.c_void_p]),
    "mi_pos_move_to_san": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_int]),
    "mi_pos_perft": (C.c_ulonglong, [C.c_void_p, C.c_int]),
    "mi_chess960_start_fen": (C.c_char_p, [C.c_int]),
    # planes
    "mi_planes_layout": (C.c_int, [C.c_int, C.c_int]),
    "mi_planes_layout_minor": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mi_planes_channels": (C.c_int, [C.c_int]),
    "mi_pos_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mi_pos_desc": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_pos_desc_for": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mi_planes_from_descs_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p]),
    "mi_planes_from_descs_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    # policy
    "mi_policy_nb_labels": (C.c_int, [C.c_int]),
    "mi_policy_nb_policy_map": (C.c_int, [C.c_int]),
    "mi_policy_label": (C.c_char_p, [C.c_int, C.c_int, C.c_int]),
    "mi_policy_flat_plane_idx": (C.c_int, [C.c_int, C.c_int]),
    "mi_pos_policy_index": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    # search
    "mi_search_default_settings": (None, [C.c_void_p]),
    "mi_search_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "mi_search_destroy": (None, [C.c_void_p]),
    "mi_search_add_position": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]),
    "mi_search_run": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_void_p]),
    "mi_search_run_timed": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_void_p]),
    "mi_search_stop": (C.c_int, [C.c_void_p]),
    "mi_search_announce_go": (C.c_int, [C.c_void_p]),
    "mi_search_cancel_go": (C.c_int, [C.c_void_p]),
    "mi_search_pv_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), c_float_p]),
    "mi_time_for_move": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mi_search_pv": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mi_search_root_children": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), c_float_p, c_float_p]),
    "mi_search_tree_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), c_float_p]),
    "mi_search_root_policy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), c_float_p]),
    "mi_search_set_active": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mi_search_reset_position": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p]),
    "mi_search_apply_move": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "mi_search_tree_fen": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "mi_search_add_lane": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_search_root_solved": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mi_search_best_move": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "mi_traindata_create": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint]),
    "mi_traindata_destroy": (None, [C.c_void_p]),
    "mi_traindata_set_phases": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mi_traindata_new_game": (C.c_int, [C.c_void_p]),
    "mi_traindata_save_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_float]),
    "mi_search_save_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mi_traindata_export_game_samples": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]),
    "mi_traindata_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mi_search_set_shared_collectors": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_search_set_adaptive_quota": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_search_set_state_budget": (C.c_int, [C.c_void_p, C.c_uint]),
    "mi_selfplay_default_settings": (None, [C.c_void_p]),
    "mi_selfplay_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p]),
    "mi_selfplay_destroy": (None, [C.c_void_p]),
    "mi_selfplay_set_start_fens": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mi_selfplay_set_epd_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mi_selfplay_play": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mi_selfplay_game": (C.c_long, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_long]),
    "mi_selfplay_get_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_policy_apply_temperature": (None, [C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_policy_get_quantile": (C.c_double, [C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_policy_apply_quantile_clipping": (None, [C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_policy_sharpen_distribution": (None, [C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_selfplay_set_phase_exporter": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mi_search_tree_dump": (C.c_long, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_long]),
    "mi_search_debug_replay": (C.c_long, [C.c_void_p, C.c_char_p, C.c_long]),
}

_lib = None


def load():
    """Loads the HIP library; raises (never falls back) when it is absent or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m crazyara_amd.build` (hipcc, gfx950). "
            "crazyara_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().mi_last_error().decode()
