"""YAML config loader with the reference's schema (``configs/config.py:4-18``, keys as in ``configs/TUM.yml`` /
``configs/UTMM.yml``) plus ``default_config`` -- the TUM.yml hot-path settings as a dict for synthetic runs."""
import copy

import yaml


def load_config(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


_TUM = {
    "dataset": "synthetic", "device": "cuda:0", "method": "vigs", "use_gt_depth": True, "white_background": False,
    "scene_radius_depth_ratio": 2, "desired_height": 480, "desired_width": 640,
    "debug": {"get_runtime_stats": False, "create_video": False, "save_keyframes": False},
    "pipeline": {"convert_SHs_python": False, "compute_cov3D_python": False, "transform_means_python": True,
                 "force_isotropic": False, "use_rgb": False},
    "tracking": {"iters": 100, "use_gt_pose": False, "dynamics_model": "const_velocity", "use_imu_loss": False,
                 "imu_T_weight": 0.0, "imu_q_weight": 0.0, "use_depth_estimate_loss": False, "pearson_weight": 0.05,
                 "position_lr": 0.001, "rotation_lr": 0.003},
    "mapping": {"iters": 150, "kf_every": 5, "niqe_kf": False, "niqe_window_size": 5, "kf_window_size": 25,
                "covisibility_level": 1, "min_covisibility": 0.95, "kf_covisibility": 0.1, "do_BA": False,
                "use_depth_estimate_loss": True, "pearson_weight": 0.05, "sh_degree": 0, "cam_t_lr": 0.001,
                "cam_q_lr": 0.003, "position_lr_init": 0.0001, "position_lr_final": 0.0000016,
                "position_lr_delay_mult": 0.01, "position_lr_max_steps": 30000, "feature_lr": 0.0025, "opacity_lr": 0.05,
                "scaling_lr": 0.001, "rotation_lr": 0.001, "rgb_lr": 0.0025, "spatial_lr_scale": 1, "percent_dense": 0.01,
                "lambda_dssim": 0.2, "min_opacity": 0.005, "densification_interval": 50, "pruning_interval": 50,
                "size_threshold": 100, "opacity_reset_interval": 500, "densify_from_iter": 0, "densify_until_iter": 50,
                "densify_grad_threshold": 0.0002},
    "cam": {"image_height": 480, "image_width": 640, "fx": 517.3, "fy": 516.5, "cx": 318.6, "cy": 255.3,
            "crop_edge": 8, "png_depth_scale": 5000.0, "fps": 30},
}


# configs/UTMM.yml's hot-path settings that differ from TUM.yml (BASELINE.json configs[2]: RGB-D + IMU): 640x330, intrinsics of the
# 1280x660 sensor scaled by 1/2 (gradslam_datasets/datautils.py:73-118), isotropic Gaussians, IMU dead-reckoning for the pose
# prediction, Pearson term in tracking, 0.002 pose learning rates, size threshold 200.
_UTMM_DELTA = {
    "desired_height": 330, "desired_width": 640,
    "pipeline": {"force_isotropic": True},
    "tracking": {"dynamics_model": "imu", "use_depth_estimate_loss": True, "pearson_weight": 0.001, "position_lr": 0.002, "rotation_lr": 0.002},
    "mapping": {"pearson_weight": 0.001, "cam_t_lr": 0.002, "cam_q_lr": 0.002, "size_threshold": 200},
    "cam": {"image_height": 330, "image_width": 640, "fx": 642.6510620117188 / 2, "fy": 641.807373046875 / 2, "cx": 654.4762573242188 / 2,
            "cy": 359.5939025878906 / 2, "png_depth_scale": 1000.0, "fps": 30},
}


def utmm_config(device="cuda:0", **overrides):
    """configs/UTMM.yml's hot-path settings on the synthetic sequence (ground-truth depth, NIQE filter off)."""
    cfg = copy.deepcopy(_TUM)
    cfg["device"] = device
    for k, v in _UTMM_DELTA.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    for k, v in overrides.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


def default_config(device="cuda:0", height=480, width=640, **overrides):
    """configs/TUM.yml's hot-path settings (iteration budgets, learning rates, pipeline flags) with ground-truth depth
    (no monocular network offline) and the NIQE keyframe filter off.  Intrinsics scale with the image size."""
    cfg = copy.deepcopy(_TUM)
    cfg["device"] = device
    sx, sy = width / 640.0, height / 480.0
    cfg["desired_height"], cfg["desired_width"] = height, width
    for k, s in (("fx", sx), ("cx", sx), ("fy", sy), ("cy", sy)):
        cfg["cam"][k] = _TUM["cam"][k] * s
    for k, v in overrides.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg
