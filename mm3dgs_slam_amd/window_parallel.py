"""Mapping-window data parallelism: one process per GPU, Gaussians replicated, each rank renders a different keyframe of
the window per optimiser step, per-Gaussian gradients are summed with one all-reduce (RCCL over xGMI on MI355X, backend
"nccl"; "gloo" in the CPU tests), then every replica takes the identical Adam step.

The reference renders ONE randomly popped keyframe per step (``slam/mapper.py:803-807``); with ``world`` ranks a step
consumes ``world`` keyframes from the same refillable stack (same seeded RNG on every rank), i.e. the optimiser sees a
window batch of ``world`` views.  ``world == 1`` is bit-identical to the reference-faithful single-view loop.

One flat fp32 buffer [P, 17 + 3(M-1) + 2] carries every parameter gradient plus the densification statistics
(||d means2D|| and visibility count), so there is exactly one sum all-reduce and one max all-reduce (radii) per step:
few, large messages -- what point-to-point xGMI links want.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

_ORDER = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_rgb")


class WindowParallel:
    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group

    def take(self, pop):
        """Pop ``world`` keyframe ids with the shared RNG; return this rank's."""
        ids = [pop() for _ in range(self.world)]
        return ids[self.rank]

    def reduce(self, gaussians, viewspace_points, visibility, radii):
        """Sum parameter gradients and densification statistics over the ranks (in place); returns
        (grad_norm_sum[P,1], visible_count[P,1], max_radii[P])."""
        P = gaussians._xyz.shape[0]
        dev = gaussians._xyz.device
        cols = []
        for name in _ORDER:
            p = getattr(gaussians, name)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            cols.append(g.reshape(P, -1))
        vs = viewspace_points.grad
        norm = torch.norm(vs[:, :2], dim=-1, keepdim=True) * visibility[:, None]
        cols += [norm, visibility[:, None].to(norm.dtype)]
        flat = torch.cat(cols, 1).contiguous()
        rmax = torch.where(visibility, radii, torch.zeros_like(radii)).to(torch.float32)
        if self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=self.group)
        off = 0
        for name in _ORDER:
            p = getattr(gaussians, name)
            n = p[0].numel() if P > 0 else 0
            if p.grad is not None:
                p.grad.copy_(flat[:, off:off + n].reshape(p.shape))
            elif n:
                p.grad = flat[:, off:off + n].reshape(p.shape).clone()
            off += n
        return flat[:, off:off + 1], flat[:, off + 1:off + 2], rmax

    def reduce_flat(self, flat, rmax=None):
        """Native-loop variant: `flat` is (a prefix of) the engine's single gradient+statistics buffer (sum), `rmax` the radii
        (max; None outside the densification phase, when neither the statistics columns nor the radii are consumed)."""
        if self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if rmax is not None:
                dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=self.group)
