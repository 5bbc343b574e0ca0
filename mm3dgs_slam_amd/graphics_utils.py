"""Camera intrinsics -> projection matrix, as ``utils/graphics_utils.py:85-94`` (``getProjectionMatrix2``)."""
import torch


def getProjectionMatrix2(znear, zfar, fx, fy, cx, cy, h, w):
    fx, fy, cx, cy, h, w = (float(v) for v in (fx, fy, cx, cy, h, w))
    depth = zfar - znear
    P = torch.zeros(4, 4)
    P[0, 0], P[0, 2] = 2 * fx / w, -(w - 2 * cx) / w
    P[1, 1], P[1, 2] = 2 * fy / h, -(h - 2 * cy) / h
    P[2, 2], P[2, 3] = zfar / depth, -(zfar * znear) / depth
    P[3, 2] = 1.0
    return P
