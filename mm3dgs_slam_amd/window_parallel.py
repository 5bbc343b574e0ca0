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
    """``world`` ranks x ``batch`` views per rank = the window batch G of one optimiser step.
    ``WindowParallel(0, 1, batch=G)`` is the single-GPU "window-batch = G" mode: the parity baseline of a G-rank run (the same
    G views per step, gradients summed locally instead of by the all-reduce)."""

    #: maps of at least this many Gaussians take the sharded optimiser step under optimizer="auto" (1 M Gaussians = 56 MB of
    #: gradients per step; at 157 k the fused Adam + projection launch of the all-reduce path is the cheaper step)
    SHARD_MIN_GAUSSIANS = 500_000

    def __init__(self, rank: int, world: int, group=None, batch: int = 1, always_reduce: bool = False, optimizer: str = "auto"):
        self.rank, self.world, self.group, self.batch = rank, world, group, int(batch)
        # optimizer: how a step's gradients become stepped parameters on every replica (native loops, fused.py):
        #   "allreduce"      -- one flat all-reduce, then the identical Adam step over the whole map on every rank;
        #   "reduce_scatter" -- reduce-scatter of the flat gradient, Adam on this rank's 1 / world of the ELEMENTS (its slice of the
        #                       parameters and of both moments), all-gather of the stepped parameters: the same bytes over the links as
        #                       the ring all-reduce moves, 1 / world of the optimiser's HBM traffic per rank;
        #   "auto"           -- "reduce_scatter" from SHARD_MIN_GAUSSIANS Gaussians on when there is more than one rank.
        assert optimizer in ("auto", "allreduce", "reduce_scatter"), optimizer
        self.optimizer = optimizer
        self.sharded_steps = 0
        # always_reduce: issue the collectives even with a single rank (an all-reduce over one rank is the identity): runs the
        # whole multi-GPU orchestration -- gradient-output loops, flat buffer, RCCL launch, separate Adam -- on a 1-GPU box
        self.always_reduce = bool(always_reduce)
        # timing (bench.py): every `sample`-th gradient all-reduce is bracketed by events on the current stream (the collective is
        # ordered after / before the work enqueued there, so the interval is the time the optimiser step waits for it)
        self.timing = False
        self._timed, self._calls, self._sample = [], 0, 8

    @property
    def _collective(self):
        return self.world > 1 or self.always_reduce

    @property
    def sharded(self):
        """True when an optimiser step goes through the gradient-output + reduce + Adam path (more than one view per step, or
        collectives forced)."""
        return self.views_per_step > 1 or self.always_reduce

    @property
    def views_per_step(self):
        return self.world * self.batch

    def take(self, pop):
        """Pop ``world * batch`` keyframe ids with the shared RNG; return this rank's ``batch`` of them (a list)."""
        ids = [pop() for _ in range(self.world * self.batch)]
        return ids[self.rank * self.batch:(self.rank + 1) * self.batch]

    @staticmethod
    def view_stats(viewspace_points, visibility, radii):
        """Densification statistics of ONE rendered view: (||d means2D|| on visible, visible as 0/1, radii on visible)."""
        norm = torch.norm(viewspace_points.grad[:, :2], dim=-1, keepdim=True) * visibility[:, None]
        return norm, visibility[:, None].to(norm.dtype), torch.where(visibility, radii, torch.zeros_like(radii)).to(torch.float32)

    @staticmethod
    def merge_stats(a, b):
        return b if a is None else (a[0] + b[0], a[1] + b[1], torch.max(a[2], b[2]))

    def reduce(self, gaussians, stats):
        """Sum the parameter gradients (``.grad``, already accumulated over this rank's views by autograd) and the
        densification statistics ``stats = (norm_sum[P,1], visible_count[P,1], max_radii[P])`` over the ranks (in place);
        returns the reduced statistics."""
        P = gaussians._xyz.shape[0]
        cols = []
        for name in _ORDER:
            p = getattr(gaussians, name)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            cols.append(g.reshape(P, -1))
        norm, count, rmax = stats
        cols += [norm, count]
        flat = torch.cat(cols, 1).contiguous()
        if self._collective:
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

    def reduce_pose_grads(self, tensors):
        """Bundle adjustment with a sharded window: each rank holds the pose gradients of ITS views only; sum them so that the
        keyframe-pose Adam step is identical on every replica.  A has-gradient flag per tensor travels with the gradients (same
        all-reduce): a pose that NO rank rendered this step keeps ``grad = None``, so Adam skips it -- moments and step counter
        untouched -- exactly like the single-rank window-batch run (and the reference's ``zero_grad(set_to_none=True)`` loop)."""
        if not self._collective or not tensors:
            return
        flat = torch.cat([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in tensors]
                         + [torch.tensor([0.0 if t.grad is None else 1.0 for t in tensors], dtype=tensors[0].dtype, device=tensors[0].device)])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        has = flat[-len(tensors):] > 0
        off = 0
        for i, t in enumerate(tensors):
            n = t.numel()
            t.grad = flat[off:off + n].reshape(t.shape).clone() if bool(has[i]) else None
            off += n

    def any_flag(self, flag: bool, device="cpu") -> bool:
        """Logical OR of a host flag over the ranks (a rank-local event such as a binning overflow must lead to the same
        decision -- re-run the loop -- on every replica)."""
        if not self._collective:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def take_all(self, pop):
        """(all `world * batch` ids of the step -- identical on every rank --, this rank's `batch` of them)."""
        ids = [pop() for _ in range(self.world * self.batch)]
        return ids, ids[self.rank * self.batch:(self.rank + 1) * self.batch]

    def reduce_small(self, t):
        """Sum of a small tensor over the ranks (bundle adjustment: the window's pose gradients), in place."""
        if self._collective:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ---- sharded optimiser step (reduce-scatter -> Adam on 1 / world of the elements -> all-gather of the parameters) ----------------
    def shard_optimizer(self, P: int) -> bool:
        if self.optimizer == "reduce_scatter":
            return True
        return self.optimizer == "auto" and self.world > 1 and P >= self.SHARD_MIN_GAUSSIANS

    def shard_bounds(self, n: int):
        """(S, lo, hi): every rank owns S consecutive elements of a flat array of n (S a multiple of 4: 16-byte aligned slices); this
        rank's are [lo, hi) -- hi clipped to n, the padding behind n belongs to the last rank(s) and is never used."""
        S = -(-n // self.world)
        S = (S + 3) // 4 * 4
        lo = min(self.rank * S, n)
        return S, lo, min(lo + S, n)

    def _has(self, name):
        if not hasattr(self, "_caps"):
            self._caps = {}
        if name not in self._caps:
            backend = dist.get_backend(self.group) if dist.is_initialized() else "none"
            # gloo (the CPU tests) has no reduce-scatter: the sum of all elements by all-reduce, then this rank's slice -- the same values
            self._caps[name] = backend != "gloo" or name != "reduce_scatter"
        return self._caps[name]

    def reduce_scatter_flat(self, flat, n, out, tail=0, rmax=None):
        """out[:S] <- this rank's slice of the element-wise sum over the ranks of flat[:n]; flat[n:n + tail] (the densification statistics)
        <- its sum over the ranks, whole, on every rank; rmax (the radii) <- its maximum.  flat must hold world * S elements (what lies
        behind n is read by the reduce-scatter and ignored); afterwards flat[:n] is unspecified."""
        S, lo, hi = self.shard_bounds(n)
        assert flat.numel() >= max(self.world * S, n + tail) and out.numel() >= S
        if not self._collective:
            out[:hi - lo].copy_(flat[lo:hi])
            return
        ev = None
        if self.timing and flat.is_cuda:
            self._calls += 1
            if self._calls % self._sample == 1:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
        if self._has("reduce_scatter"):
            dist.reduce_scatter_tensor(out[:S], flat[:self.world * S], op=dist.ReduceOp.SUM, group=self.group)
            if tail:
                dist.all_reduce(flat[n:n + tail], op=dist.ReduceOp.SUM, group=self.group)
        else:
            # (ONE all-reduce over gradients + statistics, like the all-reduce path issues it: gloo's ring cuts a buffer into chunks by its
            #  length and sums each chunk in its own rank order, so only the same span gives the same bits -- which is what lets the CPU
            #  tests hold the sharded step to the replicated one bit for bit)
            dist.all_reduce(flat[:n + tail], op=dist.ReduceOp.SUM, group=self.group)
            out[:hi - lo].copy_(flat[lo:hi])
        if rmax is not None:
            dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=self.group)
        if ev is not None:
            ev[1].record()
            self._timed.append((ev, n * flat.element_size()))
            if len(self._timed) > 256:
                del self._timed[:-256]

    def all_gather_flat(self, full, shard, n):
        """full[:n] <- the ranks' shards in rank order (shard = this rank's S elements; full holds world * S)."""
        S, lo, hi = self.shard_bounds(n)
        assert full.numel() >= self.world * S and shard.numel() >= S
        if not self._collective:
            full[lo:hi].copy_(shard[:hi - lo])
            return
        dist.all_gather_into_tensor(full[:self.world * S], shard[:S], group=self.group)

    def reduce_flat(self, flat, rmax=None):
        """Native-loop variant: `flat` is (a prefix of) the engine's single gradient+statistics buffer (sum), `rmax` the radii
        (max; None outside the densification phase, when neither the statistics columns nor the radii are consumed)."""
        if self._collective:
            ev = None
            if self.timing and flat.is_cuda:
                self._calls += 1
                if self._calls % self._sample == 1:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if rmax is not None:
                dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=self.group)
            if ev is not None:
                ev[1].record()
                self._timed.append((ev, flat.numel() * flat.element_size()))
                if len(self._timed) > 256:      # (a long run must not hoard events -- the last 256 samples are the statistic)
                    del self._timed[:-256]

    def allreduce_stats(self):
        """{calls, sampled, ms_per_call, bytes_per_call} of the gradient all-reduces since timing was switched on (synchronises)."""
        if not self._timed:
            return {"calls": self._calls, "sampled": 0, "ms_per_call": None, "bytes_per_call": None}
        torch.cuda.synchronize()
        ms = [e[0].elapsed_time(e[1]) for e, _ in self._timed]
        return {"calls": self._calls, "sampled": len(ms), "ms_per_call": sum(ms) / len(ms), "bytes_per_call": sum(b for _, b in self._timed) / len(self._timed)}
