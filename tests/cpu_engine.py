"""TEST INFRASTRUCTURE: a CPU stand-in for `mm3dgs_slam_amd.fused.FusedEngine` that executes the documented semantics of the C-ABI
loops (include/mm3dgs.h: mm3dgs_slam_track, mm3dgs_slam_map, mm3dgs_slam_forward, mm3dgs_slam_visibility) with the torch graph, the
CPU oracle rasterizer and torch arithmetic -- dereferencing the very pointers, structs and step counters `fused.py` hands to the
library (the tensors live in host memory here, so `from_address` views are what the kernels' global-memory accesses are on the GPU).

It exists so that the HOST side of the native loops (`FusedTracker` / `FusedMapper`: run grouping between pruning steps, Adam step
accounting through `Mm3dgsMapAdam.step`, statistics in pruning iterations, bundle-adjustment state, the splatam schedule, the loss
configuration) can be run on CPU against the trajectories of the reference's own classes (tests/test_golden_slam.py, fixtures G9).
The kernels themselves are checked against the same semantics on the GPU (tests/test_gpu_fused.py).  Never imported by the product."""
import ctypes as C

import numpy as np
import torch

from mm3dgs_slam_amd.loss_utils import pearson_loss, rel_pose_loss, ssim
from mm3dgs_slam_amd.renderer import Renderer
from mm3dgs_slam_amd.tracker import _FrozenMap
from oracle.raster_ref import RefRasterizer


def _view(ptr, n, ctype=C.c_float):
    """The n elements at a raw host address as a tensor sharing that memory."""
    return torch.from_numpy(np.ctypeslib.as_array((ctype * int(n)).from_address(int(ptr))))


def loss_from_config(lc, out6, gt, ref):
    """Mm3dgsLossConfig semantics (include/mm3dgs.h): {total, l1, 1-ssim, 1-rho or depth term}."""
    image, depth, sil = out6[:3], out6[3], out6[4]
    smask = sil > lc.sil_thr

    def mask_of(bits):
        m = torch.ones_like(smask)
        if bits & 1:
            m = m & smask
        if bits & 2:
            m = m & (ref > 0)
        return m
    total = image.sum() * 0.0
    if lc.w_l1 != 0:
        a = (image - gt).abs()[:, mask_of(lc.l1_mask)]
        total = total + lc.w_l1 * (a.sum() if lc.l1_sum else a.mean())
    if lc.w_ssim != 0:
        total = total + lc.w_ssim * (1.0 - ssim(image, gt))
    if lc.w_pearson != 0:
        total = total + lc.w_pearson * pearson_loss(depth, ref, mask=mask_of(lc.pearson_mask) if lc.pearson_mask else None,
                                                    invert_estimate=bool(lc.pearson_invert))
    if lc.w_depth_l1 != 0:
        d = (ref - depth).abs()[mask_of(lc.depth_l1_mask)]
        total = total + lc.w_depth_l1 * (d.sum() if lc.l1_sum else d.mean())
    return total


def _adam(p, g, m, v, t, lr, b1, b2, eps):
    """torch.optim.Adam's update (the formula of slam_preprocess_bwd_kernel / slam_pose_finish_kernel), in place."""
    m.lerp_(g, 1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1, bc2s = 1.0 - b1 ** t, (1.0 - b2 ** t) ** 0.5
    p.sub_((lr / bc1) * (m / (v.sqrt() / bc2s + eps)))


class _CpuLib:
    """The one library entry point fused.py calls directly on the engine's handle: mm3dgs_adam (multi-GPU window: the all-reduced
    gradients are stepped by a separate launch)."""

    @staticmethod
    def mm3dgs_adam(table, n_groups, step, beta1, beta2, eps, stream):
        with torch.no_grad():
            for i in range(n_groups):
                e = table[i]
                p, g, m, v = (_view(ptr, e.n) for ptr in (e.param, e.grad, e.exp_avg, e.exp_avg_sq))
                _adam(p, g, m, v, int(step), e.lr, beta1, beta2, eps)
        return 0


def install(fused_module, registry=None):
    """Route mm3dgs_slam_amd.fused to CPU engines (call under monkeypatch, or in a spawned worker): the product's eligibility rule
    minus "the device is a GPU", one CpuEngine per renderer, no HIP stream."""
    registry = {} if registry is None else registry
    real = fused_module.FusedEngine.__dict__["eligible"].__func__
    return dict(eligible=staticmethod(lambda c, g: real(dict(c, device="cuda:0"), g)),
                _engine=lambda renderer: registry.setdefault(id(renderer), CpuEngine(renderer)), _stream=lambda: None), registry


class CpuEngine:
    def headroom(self):
        return 0.0          # (always snapshot: the stand-in has no capacity model)

    lib = _CpuLib()

    def __init__(self, renderer):
        self.r = renderer
        self.cfg = renderer.cfg
        self.dev = torch.device("cpu")
        self.H, self.W = renderer.image_height, renderer.image_width
        self.R = Renderer(self.cfg, rasterizer_cls=RefRasterizer, mode="reference")
        self.out = torch.zeros(6, self.H, self.W)
        self.loss = torch.zeros(4)
        self.grads = None
        self.P = -1
        self.calls = []
        self.view_log = []

    # ---- what fused.py uses of FusedEngine --------------------------------------------------------------------------------------
    def _ensure(self, P, need_grads):
        if P != self.P:
            self.P, self.grads = P, None
        if need_grads and self.grads is None:
            self.flat = torch.zeros(16 * P + 64)
            o = [0, 3 * P, 6 * P, 7 * P, 10 * P, 14 * P, 15 * P, 16 * P]
            v = lambda i, shape: self.flat[o[i]:o[i + 1]].view(shape)
            self.grads = dict(xyz=v(0, (P, 3)), f_dc=v(1, (P, 1, 3)), opacity=v(2, (P, 1)), scaling=v(3, (P, 3)), rotation=v(4, (P, 4)))
            self.acc = torch.zeros(14 * P)                                         # window-batch mode: sum of the local views' gradients
            self.stat_delta = (torch.zeros(P), v(5, (P, 1)), v(6, (P, 1)))         # max radii | accum | denom of one optimiser step

    def check_capacity(self):
        return True

    def _render(self, pc, pose):
        res = self.R.render(pc, pose)
        return torch.cat([res["render"], res["depth"]], 0), res

    def forward(self, pose, g, need_grads=False):
        with torch.no_grad():
            self.out = self._render(g, pose)[0]

    def visibility(self, pose, g, seen):
        with torch.no_grad():
            seen += (self._render(g, pose)[1]["radii"] > 0).to(seen.dtype)

    def track_loop(self, n_iter, pose, g, lcfg, gt_color, ref, ad):
        """mm3dgs_slam_track: n iterations of {render at *ad.pose, loss, backward, pose Adam step} -- the map is not touched."""
        assert ad.pose == pose.data_ptr()
        pbuf, m, v = _view(ad.pose, 7), _view(ad.m, 7), _view(ad.v, 7)
        step = _view(ad.step, 1, C.c_int32)
        prior = _view(ad.prior_pose, 7).clone() if ad.prior_pose else None
        best = _view(ad.best, 8) if ad.best else None          # Mm3dgsPoseAdam.best: { loss, pose[7] }
        frozen = _FrozenMap(g)
        for _ in range(n_iter):
            with torch.enable_grad():          # (fused.py calls the loops under no_grad: the library needs no autograd)
                p = pbuf.clone().requires_grad_(True)
                out6, _ = self._render(frozen, p)
                loss = loss_from_config(lcfg, out6, gt_color, ref)
                if prior is not None and (ad.prior_w_t != 0 or ad.prior_w_q != 0):
                    t_l, q_l = rel_pose_loss(p, prior, safe=True)
                    loss = loss + ad.prior_w_t * t_l + ad.prior_w_q * q_l
                loss.backward()
            with torch.no_grad():
                t = int(step[0]) + 1
                step[0] = t
                _adam(pbuf[:4], p.grad[:4], m[:4], v[:4], t, ad.lr_q, ad.beta1, ad.beta2, ad.eps)
                _adam(pbuf[4:], p.grad[4:], m[4:], v[4:], t, ad.lr_t, ad.beta1, ad.beta2, ad.eps)
                if best is not None and float(loss) < float(best[0]):      # the loss at the rendered pose, the candidate = the stepped pose
                    best[0] = float(loss)
                    best[1:] = pbuf
            self.out, self.loss = out6.detach(), torch.tensor([float(loss), 0.0, 0.0, 0.0])
        self.calls.append(("track", n_iter))

    def can_adam_project(self, g):
        return int(g._xyz.shape[0]) > 0

    def adam_project(self, next_pose, g, grads, map_adam):
        """mm3dgs_slam_adam_project: the map's Adam step from the gradient arrays (opt_mask honoured); the projection of the next view it
        also launches has no counterpart here (map_loop renders from scratch)."""
        names = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation")
        P = getattr(g, names[0]).shape[0]
        keep = _view(map_adam.opt_mask, P, C.c_uint8).bool() if map_adam.opt_mask else None
        with torch.no_grad():
            for k, (n_, gname) in enumerate(zip(names, ("xyz", "f_dc", "opacity", "scaling", "rotation"))):
                p = getattr(g, n_)
                assert map_adam.param[k] == p.data_ptr(), "Mm3dgsMapAdam.param does not point at the model's tensor"
                n = p.numel()
                pv, mv, vv = _view(map_adam.param[k], n), _view(map_adam.exp_avg[k], n), _view(map_adam.exp_avg_sq[k], n)
                gk = grads[gname].reshape(P, -1)
                if keep is not None:
                    gk = gk * keep[:, None]
                _adam(pv, gk.reshape(-1), mv, vv, int(map_adam.step), map_adam.lr[k], map_adam.beta1, map_adam.beta2, map_adam.eps)
        self.calls.append(("adam_project",))
        self.pending_projection = next_pose.data_ptr()

    def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, keep_tile_order=False, want_loss=True, projected=False):
        """mm3dgs_slam_map: one iteration per view -- render, mapping loss, backward; densification statistics when given; the map's
        Adam step (state through the struct's pointers, step number map_adam.step + i) or, without it, gradient outputs; the view's
        own pose Adam (bundle adjustment) when it carries one."""
        names = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation")
        params = [getattr(g, n) for n in names]
        P = params[0].shape[0]
        # MM3DGS_FWD_PROJECTED: the caller promises that adam_project launched the projection of views[0] -- for exactly that pose buffer
        assert (not projected) or getattr(self, "pending_projection", None) == views[0][0].data_ptr(), "projected=True without a matching adam_project"
        self.pending_projection = None
        for i, view in enumerate(views):
            pose_buf, gt_color, ref = view[:3]
            pad = view[3] if len(view) > 3 else None
            dpose_out = view[4] if len(view) > 4 else None        # Mm3dgsMapView.dpose_out_or_null: the pose gradient is written out, no step
            with torch.enable_grad():
                pose = pose_buf.clone().requires_grad_(pad is not None or dpose_out is not None)
                out6, res = self._render(g, pose)
                loss = loss_from_config(lcfg, out6, gt_color, ref)
                for p in params:
                    p.grad = None
                loss.backward()
            gr = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
            with torch.no_grad():
                vis = res["visibility_filter"]
                if stats is not None:
                    max_radii2D, accum, denom = stats
                    max_radii2D[vis] = torch.max(max_radii2D[vis], res["radii"][vis].to(max_radii2D.dtype))
                    accum[vis] += torch.norm(res["viewspace_points"].grad[vis, :2], dim=-1, keepdim=True)
                    denom[vis] += 1
                if map_adam is not None:
                    keep = _view(map_adam.opt_mask, P, C.c_uint8).bool() if map_adam.opt_mask else None
                    t = int(map_adam.step) + i
                    for k, (p, gk) in enumerate(zip(params, gr)):
                        assert map_adam.param[k] == p.data_ptr(), "Mm3dgsMapAdam.param does not point at the model's tensor"
                        n = p.numel()
                        pv, mv, vv = _view(map_adam.param[k], n), _view(map_adam.exp_avg[k], n), _view(map_adam.exp_avg_sq[k], n)
                        gk = gk.reshape(P, -1)
                        if keep is not None:
                            gk = gk * keep[:, None]
                        _adam(pv, gk.reshape(-1), mv, vv, t, map_adam.lr[k], map_adam.beta1, map_adam.beta2, map_adam.eps)
                elif grads is not None:
                    for name, gk in zip(("xyz", "f_dc", "opacity", "scaling", "rotation"), gr):
                        grads[name].copy_(gk.reshape(grads[name].shape))
                if dpose_out is not None:
                    assert pad is None
                    dpose_out.copy_(pose.grad)
                if pad is not None:
                    pb, m, v, step = _view(pad.pose, 7), _view(pad.m, 7), _view(pad.v, 7), _view(pad.step, 1, C.c_int32)
                    t = int(step[0]) + 1
                    step[0] = t
                    _adam(pb[:4], pose.grad[:4], m[:4], v[:4], t, pad.lr_q, pad.beta1, pad.beta2, pad.eps)
                    _adam(pb[4:], pose.grad[4:], m[4:], v[4:], t, pad.lr_t, pad.beta1, pad.beta2, pad.eps)
                for p in params:
                    p.grad = None
            self.out = out6.detach()
        self.calls.append(("map", len(views), stats is not None, map_adam is not None))
        self.hints = getattr(self, "hints", []) + [(bool(keep_tile_order), bool(want_loss))]      # (speed hints of the C call: no effect on the results)
        self.view_log.append(tuple(round(float(v[1].double().sum()), 4) for v in views))      # which views (by their colour target) this call rendered
