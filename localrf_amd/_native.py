"""ctypes binding of liblrf_hip.so (C ABI in include/lrf.h).

There is no fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LRF_LIB", os.path.join(_HERE, "csrc", "liblrf_hip.so"))   # LRF_LIB: experiment builds

LRF_FLAG_WHITE_BG = 1
LRF_FLAG_RELU_DENS = 2
LRF_FLAG_MLP_VALU = 4
LRF_FLAG_MLP_F32 = 8
LRF_FLAG_ROWS_SAVED = 16
LRF_FLAG_SORT_RAYS = 32
LRF_FLAG_PE_OFF = 64
LRF_FLAG_PLANE_EVENTS = 128

_f = C.c_void_p  # device float*


class LrfParams(C.Structure):
    _fields_ = [("density_plane", _f * 3), ("density_line", _f * 3),
                ("app_plane", _f * 3), ("app_line", _f * 3),
                ("basis", _f), ("w1", _f), ("b1", _f), ("w2", _f), ("b2", _f), ("w3", _f), ("b3", _f),
                ("grid", C.c_int32 * 3), ("fea_pe", C.c_int32), ("view_pe", C.c_int32), ("feature_c", C.c_int32)]


class LrfField(C.Structure):
    _fields_ = [("cache", C.c_void_p), ("alpha_vol", _f), ("alpha_dim", C.c_int32 * 3),
                ("alpha_aabb", C.c_float * 6), ("aabb", C.c_float * 6), ("grid", C.c_int32 * 3),
                ("density_shift", C.c_float), ("distance_scale", C.c_float), ("weight_thres", C.c_float),
                ("term_T", C.c_float),
                ("basis", _f), ("w1", _f), ("b1", _f), ("w2", _f), ("b2", _f), ("w3", _f), ("b3", _f),
                ("fea_pe", C.c_int32), ("view_pe", C.c_int32), ("feature_c", C.c_int32)]


class LrfSceneField(C.Structure):
    _fields_ = [("field", C.POINTER(LrfField)), ("z", _f), ("S", C.c_int32), ("flags", C.c_uint32),
                ("workspace", C.c_void_p)]


class LrfGrads(C.Structure):
    _fields_ = [("density_plane", _f * 3), ("density_line", _f * 3),
                ("app_plane", _f * 3), ("app_line", _f * 3),
                ("basis", _f), ("w1", _f), ("b1", _f), ("w2", _f), ("b2", _f), ("w3", _f), ("b3", _f),
                ("zero_base", _f), ("zero_floats", C.c_int64)]


LRF_ADAM_MAX = 64
LRF_POSE_MAX = 64


class LrfAdamTensor(C.Structure):
    _fields_ = [("p", _f), ("g", _f), ("m", _f), ("v", _f), ("n", C.c_int64),
                ("step_size", C.c_float), ("bc2_sqrt", C.c_float)]


class LrfTvSeg(C.Structure):
    _fields_ = [("x", _f), ("g", _f), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("scale", C.c_float)]


class LrfFlowLoss(C.Structure):
    _fields_ = [("cam2world", _f), ("frame", C.c_void_p), ("fwd_off", C.c_void_p), ("dirs", _f), ("depth", _f), ("ij", C.c_void_p),
                ("fwd_flow", _f), ("fwd_mask", _f), ("bwd_flow", _f), ("bwd_mask", _f), ("focal", _f), ("center", _f),
                ("F", C.c_int32), ("V", C.c_int32), ("n", C.c_int32), ("quantile", C.c_float)]


LRF_LOSS_MAX_PER_VIEW = 4096
LRF_LOSS_TERMS_MAX = 8


class LrfBatchGather(C.Structure):
    _fields_ = [("images", _f), ("fwd_flow", _f), ("bwd_flow", _f), ("invdepths", _f), ("view_ids", C.c_void_p), ("pix", C.c_void_p),
                ("V", C.c_int32), ("n", C.c_int32), ("HW", C.c_int32), ("n_images", C.c_int32)]


class LrfLossTerms(C.Structure):
    _fields_ = [("x", _f * LRF_LOSS_TERMS_MAX), ("n", C.c_int32 * LRF_LOSS_TERMS_MAX), ("a", C.c_float * LRF_LOSS_TERMS_MAX),
                ("b", C.c_float * LRF_LOSS_TERMS_MAX), ("count", C.c_int32), ("s", _f)]


# every symbol include/lrf.h and include/lrf_debug.h declare: (restype, argtypes)
SYMBOLS = {
    "lrf_abi_version": (C.c_int, []),
    "lrf_last_error": (C.c_char_p, []),
    "lrf_error_slot": (C.c_void_p, []),
    "lrf_debug_set_dump": (None, [C.c_void_p]),
    "lrf_debug_set_lds_lines": (None, [C.c_int]),
    "lrf_debug_set_pipe_chunk": (None, [C.c_int]),
    "lrf_debug_set_scene_fuse": (None, [C.c_int]),
    "lrf_debug_saved_row_offset": (C.c_int64, [C.c_int, C.c_uint64, C.c_int]),
    "lrf_debug_set_bwd_overlap": (None, [C.c_int]),
    "lrf_debug_set_train_fwd_engine": (None, [C.c_int]),
    "lrf_workspace_layout_bwd": (None, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "lrf_cache_bytes": (C.c_size_t, [C.POINTER(C.c_int32)]),
    "lrf_pack_field": (C.c_int, [C.POINTER(LrfParams), C.c_void_p, C.c_void_p]),
    "lrf_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "lrf_render_fwd": (C.c_int, [C.POINTER(LrfField), _f, _f, C.c_int32, C.c_int32, C.c_uint32, C.c_float,
                                 _f, _f, _f, _f, C.c_void_p, C.c_void_p]),
    "lrf_render_fwd_profile": (C.c_int, [C.POINTER(LrfField), _f, _f, C.c_int32, C.c_int32, C.c_uint32,
                                         C.c_float, _f, _f, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "lrf_render_fwd_train": (C.c_int, [C.POINTER(LrfField), _f, _f, C.c_int32, C.c_int32, C.c_uint32, _f, _f,
                                       C.c_void_p, C.c_void_p]),
    "lrf_workspace_bytes_bwd": (C.c_size_t, [C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "lrf_workspace_bytes_bwd_cfg": (C.c_size_t, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_uint32]),
    "lrf_render_bwd": (C.c_int, [C.POINTER(LrfField), C.POINTER(LrfParams), _f, _f, C.c_int32, C.c_int32,
                                 C.c_uint32, _f, _f, C.POINTER(LrfGrads), _f, C.c_void_p, C.c_void_p]),
    "lrf_render_bwd_wait": (C.c_int, [C.c_int32, C.c_void_p]),
    "lrf_density_feature": (C.c_int, [C.POINTER(LrfField), _f, C.c_int32, _f, C.c_void_p]),
    "lrf_app_feature": (C.c_int, [C.POINTER(LrfField), _f, C.c_int32, _f, C.c_void_p]),
    "lrf_sample_ray_aabb": (C.c_int, [_f, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, _f,
                                      C.c_int32, C.c_int32, _f, _f, C.c_void_p, C.c_void_p]),
    "lrf_sample_ray_contracted": (C.c_int, [_f, _f, _f, C.c_int32, C.c_int32, _f, C.c_void_p]),
    "lrf_z_schedule": (C.c_int, [C.c_int32, _f, _f, _f, C.c_void_p]),
    "lrf_adam_step": (C.c_int, [C.POINTER(LrfAdamTensor), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "lrf_adam_step_pack": (C.c_int, [C.POINTER(LrfAdamTensor), C.c_int32, _f, C.c_float, C.c_float, C.c_float, C.POINTER(LrfParams), C.c_void_p, C.c_void_p]),
    "lrf_photo_loss_fwd": (C.c_int, [_f, _f, _f, _f, C.c_int32, _f, _f, C.c_void_p]),
    "lrf_photo_loss_bwd": (C.c_int, [_f, _f, _f, _f, _f, C.c_int32, _f, C.c_void_p]),
    "lrf_batch_gather": (C.c_int, [C.POINTER(LrfBatchGather), _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "lrf_loss_combine_fwd": (C.c_int, [C.POINTER(LrfLossTerms), _f, _f, C.c_void_p]),
    "lrf_loss_combine_bwd": (C.c_int, [_f, _f, C.c_int32, _f, C.c_void_p]),
    "lrf_rows_gather": (C.c_int, [_f, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f, C.c_void_p]),
    "lrf_rows_gather_bwd": (C.c_int, [_f, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _f, C.c_void_p]),
    "lrf_adam_step_dev": (C.c_int, [C.POINTER(LrfAdamTensor), C.c_int32, _f, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "lrf_density_l1_workspace": (C.c_size_t, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lrf_density_l1_fwd": (C.c_int, [C.POINTER(_f), C.POINTER(_f), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.c_float, C.c_int32, C.c_void_p, _f, C.c_void_p]),
    "lrf_density_l1_bwd": (C.c_int, [C.POINTER(_f), C.POINTER(_f), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.c_void_p, _f, C.POINTER(_f), C.POINTER(_f), C.c_void_p]),
    "lrf_density_l1_bwd_acc": (C.c_int, [C.POINTER(_f), C.POINTER(_f), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.c_void_p, _f, C.POINTER(_f), C.POINTER(_f), C.c_void_p]),
    "lrf_pose_assemble": (C.c_int, [C.POINTER(_f), C.POINTER(_f), C.c_int32, C.c_int32, _f, C.c_void_p]),
    "lrf_pose_assemble_bwd": (C.c_int, [C.POINTER(_f), C.c_int32, C.c_int32, _f, _f, _f, C.c_void_p]),
    "lrf_dense_alpha": (C.c_int, [C.POINTER(LrfField), _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                  C.c_uint32, _f, C.c_void_p]),
    "lrf_alpha_pool_threshold": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.c_float, _f, C.c_void_p]),
    "lrf_tv_workspace": (C.c_size_t, [C.POINTER(LrfTvSeg), C.c_int32]),
    "lrf_tv_loss_fwd": (C.c_int, [C.POINTER(LrfTvSeg), C.c_int32, C.c_float, C.c_void_p, _f, C.c_void_p]),
    "lrf_tv_loss_bwd": (C.c_int, [C.POINTER(LrfTvSeg), C.c_int32, C.c_float, _f, C.c_void_p]),
    "lrf_upsample_bilinear": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, _f, C.c_int32, C.c_int32, C.c_void_p]),
    "lrf_flow_loss_fwd": (C.c_int, [C.POINTER(LrfFlowLoss), _f, _f, C.c_void_p]),
    "lrf_flow_loss_bwd": (C.c_int, [C.POINTER(LrfFlowLoss), _f, _f, C.c_float, _f, _f, _f, _f, _f, C.c_void_p]),
    "lrf_depth_loss_fwd": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_float, _f, _f, _f, C.c_void_p]),
    "lrf_depth_loss_bwd": (C.c_int, [_f, _f, C.c_int32, C.c_int32, _f, _f, _f, C.c_float, _f, C.c_void_p]),
    "lrf_scene_rays": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f, _f, C.c_int32, _f, _f, C.c_int32,
                                 C.c_int32, C.c_int32, _f, _f, C.c_void_p, C.c_void_p]),
    "lrf_scene_rays_bwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f, C.c_int32, _f, _f, C.c_int32,
                                     C.c_int32, C.c_int32, _f, _f, _f, _f, _f, C.c_void_p]),
    "lrf_scene_blend": (C.c_int, [_f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _f, C.c_void_p]),
    "lrf_scene_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f, _f, C.c_int32, _f, _f, C.c_int32, C.c_int32,
                                C.c_int32, C.POINTER(LrfSceneField), C.c_float, C.c_int32, _f, _f,
                                _f, _f, _f, _f, C.c_void_p, _f, _f, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lrf_scene_blend_bwd": (C.c_int, [_f, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _f,
                                      C.c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"localrf_amd: {LIB_PATH} not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the render path.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if h.lrf_abi_version() != 7:
            raise NativeError("localrf_amd: ABI version mismatch")
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed: {lib().lrf_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "native call needs contiguous tensors"
    return C.c_void_p(t.data_ptr())
