"""mm3dgs_slam_amd -- MI355X-native differentiable 3D-Gaussian rasterizer + pose/map optimisation hot path
behind MM3DGS-SLAM's render boundary (reference: slam/renderer.py, slam/tracker.py, slam/mapper.py).

The compute path is the C-ABI library ``csrc/libmm3dgs_hip.so`` (hand-written HIP for gfx950).  There is no CPU or
PyTorch fallback: importing ``mm3dgs_slam_amd.rasterizer`` works without the library (so host logic can be unit
tested), but any render call raises if the library is missing or the tensors are not on a GPU.
"""
__version__ = "0.1.0"
