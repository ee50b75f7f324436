"""ctypes binding of libfyrox_hip.so (the C ABI in include/fyrox_hip.h).

There is NO CPU fallback: if the shared library is missing, or no MI355X is visible when a
context is created, this module raises.  Nothing here touches the CPU test oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FYX_LIB_PATH: another build of the SAME library (kernel experiments, tools/exp/build_variants.sh); never a fallback
LIB_PATH = os.environ.get("FYX_LIB_PATH") or os.path.join(_HERE, "libfyrox_hip.so")

FYX_OK = 0
FYX_ERR_INVALID_ARG = -1
FYX_ERR_NO_DEVICE = -2
FYX_ERR_HIP = -3
FYX_ERR_OOM = -4
FYX_ERR_UNKNOWN_ID = -5
FYX_ERR_BONE_INDEX = -6
FYX_ERR_MISSING_ATTRIBUTE = -7
FYX_ERR_UNSUPPORTED = -8

_STATUS_NAMES = {
    0: "FYX_OK", -1: "FYX_ERR_INVALID_ARG", -2: "FYX_ERR_NO_DEVICE", -3: "FYX_ERR_HIP",
    -4: "FYX_ERR_OOM", -5: "FYX_ERR_UNKNOWN_ID", -6: "FYX_ERR_BONE_INDEX",
    -7: "FYX_ERR_MISSING_ATTRIBUTE", -8: "FYX_ERR_UNSUPPORTED",
}


class SkinJob(ctypes.Structure):
    """fyx_skin_job"""
    _fields_ = [("mesh_id", c_uint64), ("d_palette", c_void_p), ("n_bones", c_uint32), ("n_instances", c_uint32),
                ("d_out_pos", c_void_p), ("d_out_normal", c_void_p), ("d_out_tangent", c_void_p)]


class SkinDesc(ctypes.Structure):
    """fyx_skin_desc (include/fyrox_hip.h)."""
    _fields_ = [("d_palette", c_void_p), ("n_bones", c_uint32), ("n_instances", c_uint32),
                ("d_blend_shape_weights", c_void_p), ("n_blend_shapes", c_uint32),
                ("d_out_pos", c_void_p), ("d_out_normal", c_void_p), ("d_out_tangent", c_void_p),
                ("d_out_vertices", c_void_p), ("out_stride", c_uint32),
                ("out_off_pos", c_int32), ("out_off_normal", c_int32), ("out_off_tangent", c_int32)]


class FyxError(RuntimeError):
    """A non-zero fyx_status returned by the native library."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.status = _STATUS_NAMES.get(code, str(code))
        super().__init__(f"{self.status}: {message}")


class NativeLibraryMissing(ImportError):
    pass


_lib = None

_P = c_void_p
_SIGS = {
    # name: (restype, [argtypes])
    "fyx_version": (c_char_p, []),
    "fyx_init": (c_int, [POINTER(c_void_p), c_int]),
    "fyx_shutdown": (None, [_P]),
    "fyx_last_error": (c_char_p, [_P]),
    "fyx_set_stream": (c_int, [_P, _P]),
    "fyx_get_stream": (c_void_p, [_P]),
    "fyx_sync": (c_int, [_P]),
    "fyx_join": (c_int, [_P]),
    "fyx_timer_begin": (c_int, [_P]),
    "fyx_timer_end": (c_int, [_P, POINTER(c_float)]),
    "fyx_set_option": (c_int, [_P, c_char_p, c_int]),
    "fyx_get_option": (c_int, [_P, c_char_p, POINTER(c_int)]),
    "fyx_debug_kernel_time": (c_int, [_P, _P, _P]),
    "fyx_debug_timeline": (c_int, [_P, _P, _P, _P, c_uint32, POINTER(c_uint32)]),
    "fyx_malloc": (c_int, [_P, c_size_t, POINTER(c_void_p)]),
    "fyx_free": (c_int, [_P, _P]),
    "fyx_malloc_streams": (c_int, [_P, c_uint32, _P, _P]),
    "fyx_memcpy_h2d": (c_int, [_P, _P, _P, c_size_t]),
    "fyx_memcpy_d2h": (c_int, [_P, _P, _P, c_size_t]),
    "fyx_mesh_upload": (c_int, [_P, c_uint64, _P, c_uint32, c_uint32, c_int, c_int, c_int, c_int, c_int]),
    "fyx_mesh_upload_soa": (c_int, [_P, c_uint64, c_uint32, _P, _P, _P, _P, _P]),
    "fyx_mesh_free": (c_int, [_P, c_uint64]),
    "fyx_mesh_info": (c_int, [_P, c_uint64, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "fyx_mesh_streams": (c_int, [_P, c_uint64] + [POINTER(c_void_p)] * 5),
    "fyx_lbs_skin": (c_int, [_P, c_uint64, _P, c_uint32, c_uint32, _P, _P, _P, _P]),
    "fyx_lbs_skin_device": (c_int, [_P, c_uint64, _P, c_uint32, c_uint32, _P, _P, _P]),
    "fyx_lbs_skin_streams": (c_int, [_P, c_uint32, _P, _P, _P, _P, _P, _P, c_uint32, c_uint32, _P, _P, _P]),
    "fyx_mesh_set_blend_shapes": (c_int, [_P, c_uint64, c_uint32, _P, c_uint32]),
    "fyx_lbs_skin_ex": (c_int, [_P, c_uint64, _P]),
    "fyx_skinned_aabb": (c_int, [_P, c_uint64, _P, c_uint32, _P]),
    "fyx_skinned_aabb_device": (c_int, [_P, c_uint64, _P, c_uint32, c_uint32, _P]),
    "fyx_calib_stream_copy": (c_int, [_P, _P, _P, c_uint32]),
    "fyx_palette": (c_int, [_P, _P, _P, c_uint32, _P]),
    "fyx_palette_device": (c_int, [_P, _P, _P, c_uint32, _P]),
    # ---- pose path -------------------------------------------------------------------------
    "fyx_init_control_only": (c_int, [POINTER(c_void_p)]),
    "fyx_tracks_data_upload": (c_int, [_P, c_uint64, c_uint32, _P, c_uint32, _P, _P, _P, _P, _P]),
    "fyx_tracks_data_free": (c_int, [_P, c_uint64]),
    "fyx_rig_create": (c_int, [_P, c_uint64, c_uint32, _P, _P, _P]),
    "fyx_rig_free": (c_int, [_P, c_uint64]),
    "fyx_bone_list_create": (c_int, [_P, c_uint64, c_uint64, c_uint32, _P]),
    "fyx_bone_list_free": (c_int, [_P, c_uint64]),
    "fyx_animator_create": (c_int, [_P, c_uint64, c_uint64, c_uint32]),
    "fyx_animator_free": (c_int, [_P, c_uint64]),
    "fyx_animator_add_animation": (c_int, [_P, c_uint64, c_uint64, _P, _P, POINTER(c_uint32)]),
    "fyx_animation_set_track_enabled": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int]),
    "fyx_animation_set_time_slice": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_float, c_float]),
    "fyx_animation_set_time_position": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_float]),
    "fyx_animation_set_speed": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_float]),
    "fyx_animation_set_loop": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int]),
    "fyx_animation_set_enabled": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int]),
    "fyx_animation_rewind": (c_int, [_P, c_uint64, c_uint32, c_uint32]),
    "fyx_animation_get_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, POINTER(c_float), POINTER(c_int), POINTER(c_int)]),
    "fyx_machine_add_parameter": (c_int, [_P, c_uint64, c_int, c_float, c_float, c_uint32, POINTER(c_uint32)]),
    "fyx_machine_set_parameter": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int, c_float, c_float, c_uint32]),
    "fyx_machine_add_layer": (c_int, [_P, c_uint64, c_float, POINTER(c_uint32)]),
    "fyx_layer_set_weight": (c_int, [_P, c_uint64, c_uint32, c_float]),
    "fyx_layer_set_mask": (c_int, [_P, c_uint64, c_uint32, _P, c_uint32]),
    "fyx_layer_add_play_animation": (c_int, [_P, c_uint64, c_uint32, c_uint32, POINTER(c_uint32)]),
    "fyx_layer_add_blend_animations": (c_int, [_P, c_uint64, c_uint32, c_uint32, _P, _P, _P, POINTER(c_uint32)]),
    "fyx_layer_add_blend_animations_by_index": (c_int, [_P, c_uint64, c_uint32, c_int32, c_uint32, _P, _P, POINTER(c_uint32)]),
    "fyx_layer_add_blend_space": (c_int, [_P, c_uint64, c_uint32, c_int32, c_uint32, _P, _P, c_uint32, _P, POINTER(c_uint32)]),
    "fyx_layer_add_state": (c_int, [_P, c_uint64, c_uint32, c_int32, POINTER(c_uint32)]),
    "fyx_layer_set_entry_state": (c_int, [_P, c_uint64, c_uint32, c_uint32]),
    "fyx_state_add_action": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int, c_int, c_uint32]),
    "fyx_layer_add_transition": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32, c_float, _P, c_uint32, POINTER(c_uint32)]),
    "fyx_layer_get_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, POINTER(c_int32), POINTER(c_int32)]),
    "fyx_machine_clear": (c_int, [_P, c_uint64]),
    "fyx_machine_get_parameter": (c_int, [_P, c_uint64, c_uint32, c_uint32, POINTER(c_int), POINTER(c_float), POINTER(c_float), POINTER(c_uint32)]),
    "fyx_layer_set_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int32, c_int32]),
    "fyx_layer_get_transition_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32, POINTER(c_float), POINTER(c_float)]),
    "fyx_layer_set_transition_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32, c_float, c_float]),
    "fyx_layer_get_node_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32, POINTER(c_int), POINTER(c_uint32), POINTER(c_float)]),
    "fyx_layer_set_node_state": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32, c_int, c_uint32, c_float]),
    "fyx_layer_reset": (c_int, [_P, c_uint64, c_uint32, c_uint32]),
    "fyx_animation_player_update": (c_int, [_P, c_uint64, c_float]),
    "fyx_absm_update": (c_int, [_P, c_uint64, c_float]),
    "fyx_animator_update_transforms": (c_int, [_P, c_uint64]),
    "fyx_animator_palette": (c_int, [_P, c_uint64, c_uint64, _P]),
    "fyx_animator_set_palette_output": (c_int, [_P, c_uint64, c_uint64, _P]),
    "fyx_animator_set_palette_output_pair": (c_int, [_P, c_uint64, c_uint64, _P, _P]),
    "fyx_animator_current_palette": (c_int, [_P, c_uint64, c_uint64, _P]),
    "fyx_debug_host_times": (c_int, [_P, _P, c_uint32]),
    "fyx_debug_frame_counter_add": (c_int, [_P, c_uint64, ctypes.c_int32]),
    "fyx_animator_set_skin_output": (c_int, [_P, c_uint64, c_uint64, c_uint64, _P, _P, _P]),
    "fyx_animator_set_local_trs": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32, _P]),
    "fyx_animator_read": (c_int, [_P, c_uint64, c_int, _P]),
    "fyx_animator_device_ptr": (c_int, [_P, c_uint64, c_int, POINTER(c_void_p)]),
    "fyx_animator_plan": (c_int, [_P, c_uint64, c_int, c_float, _P, _P, _P, _P, c_uint32, POINTER(c_uint32)]),
    # signals / events / root motion
    "fyx_animation_add_signal": (c_int, [_P, c_uint64, c_uint32, c_float, c_int, POINTER(c_uint32)]),
    "fyx_animation_set_signal_enabled": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int]),
    "fyx_animation_set_max_event_capacity": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_uint32]),
    "fyx_animation_pop_event": (c_int, [_P, c_uint64, c_uint32, c_uint32, POINTER(c_int32)]),
    "fyx_animation_event_count": (c_int, [_P, c_uint64, c_uint32, c_uint32, POINTER(c_uint32)]),
    "fyx_animation_clear_events": (c_int, [_P, c_uint64, c_uint32, c_uint32]),
    "fyx_animation_set_root_motion_settings": (c_int, [_P, c_uint64, c_uint32, c_int32, c_int, c_int, c_int, c_int]),
    "fyx_animator_track_root_motion": (c_int, [_P, c_uint64, c_int]),
    "fyx_animation_read_root_motion": (c_int, [_P, c_uint64, c_uint32, _P]),
    "fyx_absm_read_root_motion": (c_int, [_P, c_uint64, c_int32, _P]),
    "fyx_layer_pop_event": (c_int, [_P, c_uint64, c_uint32, c_uint32, _P, POINTER(c_int)]),
    "fyx_animator_property_count": (c_int, [_P, c_uint64, POINTER(c_uint32)]),
    "fyx_animator_property_slot": (c_int, [_P, c_uint64, c_int32, c_int32, POINTER(c_int32)]),
    "fyx_animator_read_properties": (c_int, [_P, c_uint64, c_int32, _P]),
    "fyx_animator_blend_shape_weights": (c_int, [_P, c_uint64, c_uint32, _P, _P, _P]),
    "fyx_layer_collect_active_animations_events": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int, _P, c_uint32, POINTER(c_uint32), _P]),
    "fyx_lbs_skin_batch": (c_int, [_P, _P, c_uint32]),
    "fyx_lbs_skin_ex_batch": (c_int, [_P, _P, _P, c_uint32]),
    "fyx_state_add_random_action": (c_int, [_P, c_uint64, c_uint32, c_uint32, c_int, _P, c_uint32]),
    "fyx_animator_set_random_seed": (c_int, [_P, c_uint64, c_uint32, c_uint64]),
    "fyx_animator_remove_animation": (c_int, [_P, c_uint64, c_uint32]),
    "fyx_scene_update": (c_int, [_P, _P, c_uint32, c_float]),
    "fyx_scene_plan": (c_int, [_P, _P, c_uint32, c_float]),
    "fyx_debug_scene_tables": (c_int, [_P, _P, c_uint32, c_int, _P, c_uint32, _P]),
    "fyx_comm_unique_id": (c_int, [_P, _P]),
    "fyx_comm_init": (c_int, [_P, _P, c_int, c_int]),
    "fyx_comm_shutdown": (c_int, [_P]),
    "fyx_allgather_f32": (c_int, [_P, _P, c_size_t, _P]),
    "fyx_shard_vertex_range": (c_int, [c_uint32, c_int, c_int, POINTER(c_uint32), POINTER(c_uint32)]),
    "fyx_debug_rig_walk": (c_int, [_P, c_uint64, _P, c_uint32, POINTER(c_uint32)]),
    "fyx_debug_rig_chunks": (c_int, [_P, c_uint64, _P, c_uint32, POINTER(c_uint32)]),
    "fyx_debug_span_value_at": (c_int, [_P, c_uint32, c_uint32, c_float, c_uint32, _P, POINTER(c_uint32)]),
    "fyx_debug_classify_fold_program": (c_int, [_P, c_uint32, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "fyx_curve_simplify": (c_int, [_P, _P, c_uint32, c_float, c_float, _P, POINTER(c_uint32)]),
    "fyx_blend_space_triangulate": (c_int, [_P, c_uint32, _P, c_uint32, POINTER(c_uint32)]),
    "fyx_shard_vertex_range_padded": (c_int, [c_uint32, c_int, c_int, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "fyx_comm_info": (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    "fyx_allgather_skinned": (c_int, [_P, c_uint32, _P, _P, _P]),
    "fyx_comm_init_all": (c_int, [_P, c_int]),
    "fyx_allgather_skinned_all": (c_int, [_P, c_int, c_uint32, _P, _P, _P]),
    "fyx_allgather_skinned_padded": (c_int, [_P, c_uint32, c_uint32, _P, _P, _P]),
    "fyx_allgather_skinned_padded_all": (c_int, [_P, c_int, c_uint32, c_uint32, _P, _P, _P]),
    "fyx_animator_plan_root_motion": (c_int, [_P, c_uint64, _P, _P, c_uint32, POINTER(c_uint32), POINTER(c_uint32), _P]),
}


def _preload_hip_runtime() -> None:
    """If torch is importable, import it first so that libfyrox_hip.so binds to the SAME
    libamdhip64.so.7 torch uses (one HIP runtime per process: device pointers and streams are
    then interchangeable between torch tensors / RCCL and this library)."""
    if os.environ.get("FYX_TEST_NO_TORCH"):   # sanitizer runs (tools/asan_gpu.sh): the system's HIP runtime, no torch
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the native library.  Raises NativeLibraryMissing when the HIP
    extension has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fyrox_amd/csrc`. fyrox_amd has no CPU fallback.")
    _preload_hip_runtime()
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(l, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def exported_symbols() -> list[str]:
    return sorted(_SIGS)
