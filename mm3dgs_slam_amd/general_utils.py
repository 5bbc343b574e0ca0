"""Small tensor helpers with the reference's conventions (``utils/general_utils.py``): ``inverse_sigmoid`` (:21-22),
``build_rotation`` (:78-99, quaternion (w,x,y,z) normalised), ``build_scaling_rotation`` (:101-110),
``strip_symmetric``/``strip_lowerdiag`` (:64-76: [xx,xy,xz,yy,yz,zz]), ``get_expon_lr_func`` (:32-62).  Device follows
the inputs (the reference hard-codes "cuda")."""
import numpy as np
import torch

from .pose_utils import quad2rotation


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def build_rotation(r):
    return quad2rotation(r)


def build_scaling_rotation(s, r):
    return build_rotation(r) * s[:, None, :]          # R @ diag(s)


def strip_lowerdiag(L):
    return torch.stack([L[:, 0, 0], L[:, 0, 1], L[:, 0, 2], L[:, 1, 1], L[:, 1, 2], L[:, 2, 2]], 1)


strip_symmetric = strip_lowerdiag


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final over max_steps with an optional sine warm-up."""
    def lr_at(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        warm = 1.0
        if lr_delay_steps > 0:
            warm = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return warm * np.exp((1 - t) * np.log(lr_init) + t * np.log(lr_final))
    return lr_at
