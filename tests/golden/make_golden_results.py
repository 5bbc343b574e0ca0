"""G10: the harness outputs of the reference (build container only).

    python tests/golden/make_golden_results.py

`slam/SLAM.py::SLAM.save_results` (:294-373) is executed AS IS on a stand-in for `self` that carries exactly the attributes it reads
(`cfg`, `estimate_pose_list`, `gt_pose_list`, `mapper.keyframes` built from the reference's own `KeyFrame`, the tracker / mapper timing
counters, `evaluate_images` returning seeded lists -- its renders need the absent CUDA extension), and the `results.npz` it writes is read
back with `np.load(..., allow_pickle=True)` the way `slam/mapper.py:65-71` reads it.  Stored: the inputs, the key list in file order,
every array, the keyframe dict fields, and `utils/eval_utils.py::evaluate_ate_rmse(..., "umeyama")` (:231-294) on three more seeded
trajectories (similarity-transformed + noisy estimates).  tests/test_golden_host.py holds this repository's `SLAM.save_results` /
`eval_utils.evaluate_ate_rmse` to it."""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402

mg.stub_modules()
for name in ("matplotlib", "matplotlib.pyplot", "gradslam_datasets", "lpipsPyTorch", "utils.depth_utils"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["gradslam_datasets"].ReplicaDataset = sys.modules["gradslam_datasets"].TUMDataset = sys.modules["gradslam_datasets"].UTMMDataset = object
sys.modules["lpipsPyTorch"].lpips = lambda *a, **k: None
_d = sys.modules["utils.depth_utils"]
_d.depth_to_rgb = _d.get_dpt = _d.get_scale_shift = None


def trajectories(seed, n):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.tensor([1.0, 0, 0, 0]) + 0.2 * torch.randn(n, 4, generator=g).cumsum(0) / n, dim=1)
    t = (0.05 * torch.randn(n, 3, generator=g)).cumsum(0)
    gt = torch.cat([q, t], 1)
    # estimate = a similarity transform of the truth + noise (what the Umeyama alignment has to undo)
    a = 0.3 * torch.randn(3, generator=g)
    A = torch.linalg.matrix_exp(torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]))
    s = 1.0 + 0.2 * float(torch.rand(1, generator=g))
    et = (s * (A @ t.T)).T + torch.tensor([0.3, -0.2, 0.1]) + 0.01 * torch.randn(n, 3, generator=g)
    eq = torch.nn.functional.normalize(q + 0.01 * torch.randn(n, 4, generator=g), dim=1)
    return torch.cat([eq, et], 1).float(), gt.float()


def main():
    with mg._CpuMode():
        from slam import SLAM as S
        from slam.mapper import KeyFrame
        from utils.eval_utils import evaluate_ate_rmse
        out = {}
        for k, (seed, n) in enumerate(((1, 12), (2, 40), (3, 5))):
            est, gt = trajectories(seed, n)
            aligned, rmse = evaluate_ate_rmse(est, gt, method="umeyama")
            out[f"ate{k}_est"], out[f"ate{k}_gt"] = est.numpy(), gt.numpy()
            out[f"ate{k}_aligned"], out[f"ate{k}_rmse"] = np.asarray(aligned), np.float64(rmse)
            # the rigid alignment of Horn (utils/eval_utils.py:193-228,249-266).  `align` calls `numpy.linalg.linalg.svd`, an alias that numpy 2
            # no longer has: bound back to numpy.linalg for the call (the arithmetic is the reference's own)
            import numpy.linalg as _la
            if not hasattr(_la, "linalg"):
                _la.linalg = _la
            aligned_h, rmse_h = evaluate_ate_rmse(est, gt, method="horn")
            out[f"ate{k}_horn_aligned"], out[f"ate{k}_horn_rmse"] = np.asarray(aligned_h), np.float64(rmse_h)
        # save_results on a stand-in self
        est, gt = trajectories(7, 9)
        last_idx = 7                                     # the reference truncates both pose lists to the frames processed
        g = torch.Generator().manual_seed(11)
        kfs = [KeyFrame(i, torch.rand(3, 6, 8, generator=g), est[i].clone(), torch.rand(6, 8, generator=g) + 1.0, None) for i in (0, 3, 5)]
        ev = ([np.float32(20.0 + i) for i in range(3)], [np.float32(0.8 + 0.01 * i) for i in range(3)], [np.float32(0.2 - 0.01 * i) for i in range(3)])
        with tempfile.TemporaryDirectory() as tmp:
            fake = types.SimpleNamespace(
                cfg={"outputdir": tmp, "debug": {"create_video": False, "get_runtime_stats": True}},
                estimate_pose_list=est.clone(), gt_pose_list=gt.clone(), mapper=types.SimpleNamespace(keyframes=kfs, mapping_time_sum=1.5, mapping_iter_count=300),
                tracker=types.SimpleNamespace(tracking_time_sum=0.5, tracking_iter_count=200), evaluate_images=lambda last: ev)
            S.SLAM.save_results(fake, last_idx)
            res = np.load(os.path.join(tmp, "results.npz"), allow_pickle=True)
            out["keys"] = np.array(list(res.keys()))
            out["sr_est"], out["sr_gt"], out["sr_last_idx"] = est.numpy(), gt.numpy(), np.int64(last_idx)
            for k in ("pose_est", "pose_gt", "ate_rmse", "psnr_list", "ssim_list", "lpips_list", "avg_tracking_it_time", "avg_mapping_it_time"):
                out["sr_" + k] = np.asarray(res[k])
            kf_read = list(res["keyframes"])
            out["sr_kf_keys"] = np.array(sorted(kf_read[0].keys()))
            out["sr_kf_idx"] = np.array([kf["idx"] for kf in kf_read])
            out["sr_kf_gt_color"] = np.stack([np.asarray(kf["gt_color"]) for kf in kf_read])
            out["sr_kf_est_pose"] = np.stack([np.asarray(kf["est_pose"]) for kf in kf_read])
            out["sr_kf_gt_depth"] = np.stack([np.asarray(kf["gt_depth"]) for kf in kf_read])
            out["sr_kf_est_depth_is_none"] = np.array([kf["est_depth"] is None for kf in kf_read])
            out["sr_timing_inputs"] = np.array([0.5, 200, 1.5, 300])
            out["sr_eval_lists"] = np.array(ev)
    path = os.path.join(HERE, "g10_results.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "bytes; keys of the reference's results.npz:", list(out["keys"]), "ate_rmse", out["sr_ate_rmse"])


if __name__ == "__main__":
    main()
