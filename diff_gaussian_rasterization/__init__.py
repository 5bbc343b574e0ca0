"""Drop-in for the reference's un-vendored ``diff_gaussian_rasterization`` extension
(/root/reference/.gitmodules:1-3; imported at slam/renderer.py:15-18).  With this repository's root on ``sys.path`` the
reference's ``slam/renderer.py`` imports these two names unchanged and runs on the MI355X-native library."""
from mm3dgs_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
