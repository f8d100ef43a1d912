"""ctypes binding of libgtx.so (C ABI declared in include/gtx.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, a
GtxError is raised. Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("GTX_LIB", _HERE / "libgtx.so"))


class GtxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libgtx error {code}: {msg}")
        self.code = code


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "n", "h", "w", "cin", "cout", "ksize", "stride", "act",
        "in_cstride", "in_coff", "out_cstride", "out_coff", "has_residual")]


class DetConfig(C.Structure):
    _fields_ = [
        ("imgsz", C.c_int), ("conf", C.c_float), ("iou", C.c_float), ("max_det", C.c_int),
        ("agnostic_nms", C.c_int), ("half", C.c_int), ("rect", C.c_int), ("nc", C.c_int),
        ("n_classes", C.c_int), ("classes", C.c_int * 80), ("max_batch", C.c_int),
        ("frame_h", C.c_int), ("frame_w", C.c_int), ("fp32_split", C.c_int), ("obj_feats", C.c_int), ("arch", C.c_int),
    ]


class TrackerConfig(C.Structure):
    _fields_ = [
        ("type", C.c_int), ("track_high_thresh", C.c_float), ("track_low_thresh", C.c_float),
        ("new_track_thresh", C.c_float), ("track_buffer", C.c_int), ("match_thresh", C.c_float),
        ("fuse_score", C.c_int), ("frame_rate", C.c_int),
        ("delta_t", C.c_int), ("inertia", C.c_float), ("use_byte", C.c_int), ("min_hits", C.c_int),
        ("reset_velocity_offset_occ", C.c_int), ("reset_pos_offset_occ", C.c_int), ("enlarge_bbox_occ", C.c_float),
        ("dampen_motion_occ", C.c_float), ("active_occ_to_lost_thresh", C.c_int), ("occ_cover_thresh", C.c_float),
        ("occ_reappear_window", C.c_int), ("init_iou_suppress", C.c_float),
        ("with_reid", C.c_int), ("proximity_thresh", C.c_float), ("appearance_thresh", C.c_float),
        ("lost_match_thr", C.c_float), ("iou_weight", C.c_float), ("reid_weight", C.c_float), ("conf_weight", C.c_float),
        ("angle_weight", C.c_float), ("penalty_p", C.c_float), ("penalty_q", C.c_float), ("reduce_step", C.c_float),
        ("tai_thr", C.c_float), ("min_track_len", C.c_int), ("alpha_fixed_emb", C.c_float),
    ]


class RegConfig(C.Structure):
    """gtx_reg_config (include/gtx.h)."""
    _fields_ = [("max_features", C.c_int), ("filter_ratio", C.c_float), ("ransac_threshold", C.c_float), ("ransac_max_iter", C.c_int),
                ("ransac_confidence", C.c_float), ("rsift_eps", C.c_float), ("seed", C.c_int)]


class GeorefChain(C.Structure):
    """gtx_georef_chain (include/gtx.h)."""
    _fields_ = [("H", C.c_double * 9), ("ortho", C.c_double * 6), ("projected", C.c_int), ("semi_major", C.c_double),
                ("flattening", C.c_double), ("lon0_deg", C.c_double), ("k0", C.c_double), ("false_easting", C.c_double),
                ("false_northing", C.c_double)]


class StabConfig(C.Structure):
    _fields_ = [
        ("downsample_ratio", C.c_float), ("max_features", C.c_int), ("ref_multiplier", C.c_float),
        ("filter_ratio", C.c_float), ("ransac_threshold", C.c_float), ("ransac_max_iter", C.c_int),
        ("ransac_confidence", C.c_float), ("mask_use", C.c_int), ("mask_margin_ratio", C.c_float),
        ("fast_threshold", C.c_int), ("n_levels", C.c_int), ("scale_factor", C.c_float),
        ("seed", C.c_uint32), ("frame_h", C.c_int), ("frame_w", C.c_int), ("clahe", C.c_int), ("affine", C.c_int), ("filter_type", C.c_int),
    ]


# name -> (restype, argtypes); kept in one table so tests can check the export list against
# include/gtx.h.
ABI_VERSION = 10       # GTX_ABI_VERSION of include/gtx.h
_P = C.c_void_p
_SIGNATURES = {
    "gtx_abi_version": (C.c_int, []),
    "gtx_last_error": (C.c_char_p, []),
    "gtx_device_count": (C.c_int, []),
    "gtx_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gtx_ctx_create_prio": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "gtx_ctx_destroy": (None, [_P]),
    "gtx_ctx_synchronize": (C.c_int, [_P]),
    "gtx_device_open_null_stream": (C.c_int, [C.c_int]),
    "gtx_device_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "gtx_streams_overlap": (C.c_int, [_P, _P, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gtx_write_table_f32": (C.c_int, [C.c_char_p, _P, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "gtx_write_table_f64": (C.c_int, [C.c_char_p, _P, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "gtx_write_csv": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, _P, _P, _P, _P, C.c_int64, C.c_int]),
    "gtx_track_anchor_walk": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, _P, _P, _P]),
    "gtx_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "gtx_dev_free": (C.c_int, [_P, _P]),
    "gtx_dev_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "gtx_dev_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "gtx_op_conv2d": (C.c_int, [_P, C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "gtx_op_conv2d_time": (C.c_int, [_P, C.POINTER(ConvDesc), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "gtx_op_conv_xcd_ranges": (C.c_int, [C.c_int, _P, _P, _P, _P]),
    "gtx_op_sppf_pool": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gtx_op_upsample2x": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int]),
    "gtx_gmc_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gtx_gmc_destroy": (None, [_P]),
    "gtx_gmc_reset": (C.c_int, [_P]),
    "gtx_gmc_apply": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(C.c_int), _P]),
    "gtx_gmc_submit_gray_dev": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "gtx_gmc_restart": (C.c_int, [_P]),
    "gtx_gmc_submit_frame_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "gtx_gmc_collect": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P]),
    "gtx_gmc_points": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), _P, _P]),
    "gtx_ecc_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(_P)]),
    "gtx_ecc_destroy": (None, [_P]),
    "gtx_ecc_reset": (C.c_int, [_P]),
    "gtx_ecc_replace_template": (C.c_int, [_P, C.c_int]),
    "gtx_ecc_exact_positions": (C.c_int, [_P, C.c_int]),
    "gtx_ecc_submit": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "gtx_ecc_submit_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "gtx_ecc_collect": (C.c_int, [_P, _P, _P, C.POINTER(C.c_double)]),
    "gtx_ecc_image": (C.c_int, [_P, C.c_int, _P]),
    "gtx_register_images": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.POINTER(C.c_int), _P, _P]),
    "gtx_sift_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "gtx_sift_destroy": (None, [_P]),
    "gtx_sift_detect": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int), _P, _P, _P]),
    "gtx_sift_stage_ms": (C.c_int, [_P, _P]),
    "gtx_sift_pyramid": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_op_match_2nn": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "gtx_op_preprocess": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int]),
    "gtx_detector_create": (C.c_int, [_P, C.POINTER(DetConfig), C.POINTER(_P)]),
    "gtx_detector_destroy": (None, [_P]),
    "gtx_detector_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64)]),
    "gtx_detector_finalize": (C.c_int, [_P]),
    "gtx_detector_input_size": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_detector_detect": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "gtx_detector_detect_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "gtx_detector_detect_batch_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "gtx_detector_submit_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "gtx_detector_collect": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "gtx_detector_gray": (_P, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_detector_raw_output": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int)]),
    "gtx_detector_raw_logits": (C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_int)]),
    "gtx_detector_layer_output": (C.c_int, [_P, C.c_int, C.c_char_p, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_detector_saturated": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "gtx_detector_fell_back": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "gtx_detector_pad_skip": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_detector_sparse_box": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_detector_features": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gtx_detector_trace": (C.c_int, [_P, C.c_int]),
    "gtx_detector_profile": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(C.c_int)]),
    "gtx_tracker_create": (C.c_int, [C.POINTER(TrackerConfig), C.POINTER(_P)]),
    "gtx_tracker_destroy": (None, [_P]),
    "gtx_tracker_reset": (C.c_int, [_P]),
    "gtx_tracker_update": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_int), _P, _P, _P, _P, _P]),
    "gtx_tracker_update_feats": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int), _P, _P, _P, _P, _P]),
    "gtx_tracker_replay": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "gtx_op_linear_assignment": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, _P, _P]),
    "gtx_stabilizer_create": (C.c_int, [_P, C.POINTER(StabConfig), C.POINTER(_P)]),
    "gtx_op_clahe": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "gtx_stabilizer_destroy": (None, [_P]),
    "gtx_stabilizer_set_ref_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int]),
    "gtx_stabilizer_set_ref_gray_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int]),
    "gtx_stabilizer_stabilize": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.POINTER(C.c_int), _P]),
    "gtx_stabilizer_stabilize_gray_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P, C.POINTER(C.c_int), _P]),
    "gtx_stabilizer_submit_gray_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int]),
    "gtx_stabilizer_collect": (C.c_int, [_P, _P, C.POINTER(C.c_int), _P]),
    "gtx_stabilizer_promote_cur": (C.c_int, [_P]),
    "gtx_stabilizer_keypoints": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), _P, _P, _P, _P]),
    "gtx_stabilizer_matches": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _P, _P, _P]),
    "gtx_stabilizer_pattern": (C.c_int, [_P, _P]),
    "gtx_stabilizer_last_ms": (C.c_int, [_P, _P]),
    "gtx_warp_boxes": (C.c_int, [_P, _P, C.c_int, _P]),
    "gtx_perspective_points": (C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    "gtx_op_estimate_affine_partial": (C.c_int, [_P, _P, C.c_int, C.c_uint, _P, _P, _P]),
    "gtx_op_georef_points": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "gtx_warp_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "gtx_warp_frame_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "gtx_yuv420_to_bgr_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "gtx_feeder_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gtx_feeder_create_on": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gtx_feeder_destroy": (None, [_P]),
    "gtx_feeder_open_file": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.c_int]),
    "gtx_feeder_open_memory": (C.c_int, [_P, _P, C.c_int64, C.c_int]),
    "gtx_feeder_open_push": (C.c_int, [_P]),
    "gtx_feeder_push": (C.c_int, [_P, _P, C.c_size_t]),
    "gtx_feeder_push_at": (C.c_int, [_P, C.c_int64, _P, C.c_size_t]),
    "gtx_feeder_finish": (C.c_int, [_P]),
    "gtx_feeder_stop": (C.c_int, [_P]),
    "gtx_feeder_next": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "gtx_feeder_wait": (C.c_int, [_P, C.c_int64, _P]),
    "gtx_feeder_release": (C.c_int, [_P, C.c_int64]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libgtx.so; raises if it has not been built (`make -C geo-trax_amd`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise GtxError(-100, f"{LIB_PATH} not found; build it with `make -C geo-trax_amd` "
                             "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gtx_abi_version() != ABI_VERSION:       # the config structs below are laid out for exactly this version of include/gtx.h
        raise GtxError(-101, f"{LIB_PATH} reports ABI version {lib.gtx_abi_version()}, these bindings are written for {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        raise GtxError(status, load().gtx_last_error().decode("utf-8", "replace"))


def ptr(a: np.ndarray | None):
    """Pointer to a C-contiguous numpy array (or NULL)."""
    if a is None:
        return None
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array passed to libgtx must be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """A device + one HIP stream (include/gtx.h: gtx_ctx_create[_prio])."""

    def __init__(self, device: int = 0, high_priority: bool | int = False):
        """high_priority: True / 1 = the device's highest stream priority, -1 = its lowest, False / 0 = default."""
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.gtx_ctx_create_prio(device, int(high_priority), C.byref(h)))
        self.handle = h
        self.device = device

    def synchronize(self):
        check(self.lib.gtx_ctx_synchronize(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gtx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # raw device memory (inputs kept resident in HBM by bench.py)
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(self.lib.gtx_dev_alloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, dptr: int):
        check(self.lib.gtx_dev_free(self.handle, C.c_void_p(dptr)))

    def dev_upload(self, dptr: int, a: np.ndarray):
        check(self.lib.gtx_dev_upload(self.handle, C.c_void_p(dptr), ptr(a), a.nbytes))

    def dev_download(self, a: np.ndarray, dptr: int):
        check(self.lib.gtx_dev_download(self.handle, ptr(a), C.c_void_p(dptr), a.nbytes))


_default_ctx: dict[int, Context] = {}


def default_device() -> int:
    """GPU of this process: $GTX_DEVICE (set per rank by the batch launcher), else 0."""
    return int(os.environ.get("GTX_DEVICE", "0") or 0)


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = default_device()
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
