"""ctypes binding of libmm3dgs_hip.so (C ABI declared in include/mm3dgs.h).  No torch types cross this boundary:
only raw device pointers, sizes and the HIP stream handle."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM3DGS_LIB: developer switch -- a variant build of the same library (tools/build_variant.sh), for accuracy / speed experiments
LIB_PATH = os.environ.get("MM3DGS_LIB") or os.path.join(_HERE, "csrc", "libmm3dgs_hip.so")


class Mm3dgsCamera(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
    ]


# flags of the SLAM entry points (include/mm3dgs.h: MM3DGS_FWD_*; tests/test_cabi.py holds these to the header)
FWD_STATE_CLEAN, FWD_SHORT_LISTS, FWD_DIRECT_BINS, FWD_KEEP_TILE_ORDER, FWD_PROJECTED = 1, 2, 4, 8, 16


class Mm3dgsHeader(C.Structure):
    _fields_ = [("num_rendered", C.c_uint32), ("overflow", C.c_uint32), ("max_tile_len", C.c_uint32),
                ("max_num_rendered", C.c_uint32), ("fwd_wave_iters", C.c_uint32), ("bwd_wave_iters", C.c_uint32),
                ("bwd_wave_visits", C.c_uint32), ("bin_cap", C.c_uint32), ("max_group_records", C.c_uint32), ("tile_order_tiles", C.c_uint32),
                ("overflow_seen", C.c_uint32), ("mean_wave_steps", C.c_uint32)]


class Mm3dgsSlamInputs(C.Structure):
    _fields_ = [("pose", C.c_void_p), ("xyz", C.c_void_p), ("f_dc", C.c_void_p), ("opacity", C.c_void_p),
                ("scaling", C.c_void_p), ("rotation", C.c_void_p), ("isotropic", C.c_int32), ("world_means", C.c_int32),
                ("f_rest", C.c_void_p), ("sh_degree", C.c_int32), ("n_rest", C.c_int32)]


class Mm3dgsSlamGrads(C.Structure):
    _fields_ = [("d_xyz", C.c_void_p), ("d_f_dc", C.c_void_p), ("d_opacity", C.c_void_p), ("d_scaling", C.c_void_p),
                ("d_rotation", C.c_void_p), ("max_radii2D", C.c_void_p), ("grad_accum", C.c_void_p), ("denom", C.c_void_p),
                ("d_f_rest", C.c_void_p)]


class Mm3dgsMapAdam(C.Structure):
    _fields_ = [("param", C.c_void_p * 5), ("exp_avg", C.c_void_p * 5), ("exp_avg_sq", C.c_void_p * 5), ("lr", C.c_double * 5),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int32), ("opt_mask", C.c_void_p),
                ("rest_param", C.c_void_p), ("rest_exp_avg", C.c_void_p), ("rest_exp_avg_sq", C.c_void_p), ("rest_lr", C.c_double)]


class Mm3dgsPoseAdam(C.Structure):
    _fields_ = [("pose", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p), ("lr_q", C.c_double),
                ("lr_t", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("prior_pose", C.c_void_p), ("prior_w_t", C.c_float), ("prior_w_q", C.c_float), ("best", C.c_void_p)]


class Mm3dgsMapView(C.Structure):
    _fields_ = [("pose", C.c_void_p), ("gt_color", C.c_void_p), ("ref_depth_or_null", C.c_void_p), ("pose_adam_or_null", C.c_void_p),
                ("dpose_out_or_null", C.c_void_p)]


class Mm3dgsLossConfig(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("w_l1", C.c_float), ("w_ssim", C.c_float), ("w_pearson", C.c_float),
                ("l1_mask", C.c_int32), ("pearson_mask", C.c_int32), ("pearson_invert", C.c_int32), ("sil_thr", C.c_float),
                ("window", C.c_float * 11), ("w_depth_l1", C.c_float), ("depth_l1_mask", C.c_int32), ("l1_sum", C.c_int32)]


class Mm3dgsAdamGroup(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_uint64), ("lr", C.c_double)]


class Mm3dgsCompactArray(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("width", C.c_int32)]


class Mm3dgsSeedOutputs(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("f_dc", C.c_void_p), ("f_rest", C.c_void_p), ("opacity", C.c_void_p), ("scaling", C.c_void_p),
                ("rotation", C.c_void_p), ("rgb", C.c_void_p)]


_P = C.c_void_p
_SIGS = {
    "mm3dgs_profile_event_overhead_ms": (C.c_double, [_P]),
    "mm3dgs_propagate_const_vel": (C.c_int, [_P, _P, _P, _P]),
    "mm3dgs_covisibility_ratio": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "mm3dgs_prune_mask": (C.c_int, [C.c_int, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P]),
    "mm3dgs_compact_work_bytes": (C.c_size_t, [C.c_size_t]),
    "mm3dgs_compact_plan": (C.c_int, [C.c_size_t, _P, _P, _P, _P]),
    "mm3dgs_compact_rows": (C.c_int, [C.c_size_t, _P, _P, C.POINTER(Mm3dgsCompactArray), C.c_int, _P]),
    "mm3dgs_seed_gaussians": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                        C.POINTER(Mm3dgsSeedOutputs), C.c_int, _P]),
    "mm3dgs_geom_bytes": (C.c_size_t, [C.c_int]),
    "mm3dgs_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mm3dgs_binning_bytes": (C.c_size_t, [C.c_size_t]),
    "mm3dgs_backward_scratch_bytes": (C.c_size_t, [C.c_int, C.c_size_t]),
    "mm3dgs_forward_geom": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.c_int, C.c_int] + [_P] * 12),
    "mm3dgs_forward_raster": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P, _P]),
    "mm3dgs_forward": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.c_int, C.c_int] + [_P] * 12 + [C.c_size_t, _P]),
    "mm3dgs_backward": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.c_int, C.c_int] + [_P] * 11 + [C.c_size_t]
                        + [_P] * 13 + [C.c_int, _P]),
    "mm3dgs_mark_visible": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, _P, _P, _P]),
    "mm3dgs_slam_forward": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.POINTER(Mm3dgsSlamInputs), _P, _P, _P, _P, _P, C.c_size_t,
                                      C.c_int, _P]),
    "mm3dgs_slam_backward": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.POINTER(Mm3dgsSlamInputs), _P, _P, _P, _P, C.c_size_t, _P, _P,
                                       C.POINTER(Mm3dgsSlamGrads), _P, C.POINTER(Mm3dgsPoseAdam), C.POINTER(Mm3dgsMapAdam), C.c_int, _P]),
    "mm3dgs_slam_visibility": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.POINTER(Mm3dgsSlamInputs), _P, _P, _P, _P]),
    "mm3dgs_slam_track": (C.c_int, [C.c_int, C.POINTER(Mm3dgsCamera), C.c_int, C.POINTER(Mm3dgsSlamInputs), _P, _P, _P, _P, _P, C.c_size_t,
                                    C.c_int, C.POINTER(Mm3dgsLossConfig), _P, _P, _P, _P, _P, _P, C.POINTER(Mm3dgsPoseAdam), _P]),
    "mm3dgs_slam_map": (C.c_int, [C.c_int, C.POINTER(Mm3dgsMapView), C.POINTER(Mm3dgsCamera), C.c_int, C.POINTER(Mm3dgsSlamInputs), _P, _P, _P,
                                  _P, _P, C.c_size_t, C.c_int, C.POINTER(Mm3dgsLossConfig), _P, _P, _P, _P, C.POINTER(Mm3dgsSlamGrads),
                                  C.POINTER(Mm3dgsMapAdam), _P]),
    "mm3dgs_loss_work_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mm3dgs_loss": (C.c_int, [C.POINTER(Mm3dgsLossConfig), _P, _P, _P, _P, _P, _P, _P]),
    "mm3dgs_adam": (C.c_int, [C.POINTER(Mm3dgsAdamGroup), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _P]),
    "mm3dgs_slam_direct_bins": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.c_size_t, C.c_int]),
    "mm3dgs_slam_adam_project": (C.c_int, [C.POINTER(Mm3dgsCamera), C.c_int, C.POINTER(Mm3dgsSlamInputs), C.POINTER(Mm3dgsSlamGrads), C.POINTER(Mm3dgsMapAdam),
                                           _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "mm3dgs_profile_enable": (None, [C.c_int]),
    "mm3dgs_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "mm3dgs_last_error": (C.c_char_p, []),
    "mm3dgs_version": (C.c_int, []),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load the shared library (once).  Raises RuntimeError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libmm3dgs_hip.so not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C {os.path.dirname(LIB_PATH)}`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


PROF_KERNELS = ("preprocess_fwd", "scan", "bin_sort", "composite_fwd", "composite_bwd", "preprocess_bwd", "loss", "adam", "composite_bwd_track", "track_fwd_bwd")


def profile_enable(mode):
    """0/False off, 1/True every kernel, 2 only the compositors (forward, backward mapping / generic, backward tracking), every 64th launch."""
    load().mm3dgs_profile_enable(int(mode))


def profile_read():
    """{kernel: (launches, total_ms)} since the previous read (waits for the recorded events)."""
    lib = load()
    out = {}
    for k, name in enumerate(PROF_KERNELS):
        n, ms = C.c_uint64(0), C.c_double(0.0)
        check(lib.mm3dgs_profile_read(k, C.byref(n), C.byref(ms)))
        out[name] = (int(n.value), float(ms.value))
    return out


def check(rc: int):
    if rc != 0:
        msg = load().mm3dgs_last_error()
        raise RuntimeError(f"mm3dgs error {rc}: {msg.decode() if msg else '?'}")
