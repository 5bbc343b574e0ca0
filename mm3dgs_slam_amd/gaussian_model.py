"""Scene state: the Gaussian map and its Adam optimiser (reference ``slam/gaussian_model.py``).

Same parameter set, activations, optimiser groups and group surgery as the reference so that the mapper loop reads the
same (file:line are into ``slam/gaussian_model.py``):
  parameters  _xyz[P,3] _features_dc[P,1,3] _features_rest[P,M-1,3] _opacity[P,1] _scaling[P,3] _rotation[P,4] _rgb[P,3]
  getters     exp / normalize / sigmoid / cat                                  (:108-137)
  optimiser   Adam(lr=0, eps=1e-15), one group per parameter, names xyz f_dc f_rest opacity scaling rotation rgb (:143-195)
  surgery     densification_postfix -> cat_tensors_to_optimizer (:418-487), prune -> prune_points -> _prune_optimizer
              (:380-417, :574-588): parameters are REPLACED by fresh ``nn.Parameter`` objects, so a gradient computed
              before the prune never reaches ``optimizer.step()`` (SURVEY.md section 3.3 -- reproduced on purpose)
  stats       max_radii2D, xyz_gradient_accum, denom; add_densification_stats (:594-598)
PLY import/export keeps the reference's attribute order (:205-257) with a dependency-free binary writer.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .general_utils import build_scaling_rotation, get_expon_lr_func, inverse_sigmoid, strip_symmetric

_GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "rgb")


class GaussianModel:
    def __init__(self, cfg):
        self.cfg = cfg
        dev = cfg["device"]
        self.active_sh_degree = 0
        self.max_sh_degree = cfg["mapping"]["sh_degree"]
        n_rest = (self.max_sh_degree + 1) ** 2 - 1
        self._xyz = torch.empty(0, 3, device=dev)
        self._features_dc = torch.empty(0, 1, 3, device=dev)
        self._features_rest = torch.empty(0, n_rest, 3, device=dev)
        self._scaling = torch.empty(0, 3, device=dev)
        self._rotation = torch.empty(0, 4, device=dev)
        self._opacity = torch.empty(0, 1, device=dev)
        self._rgb = torch.empty(0, 3, device=dev)
        self.max_radii2D = torch.empty(0, device=dev)
        self.xyz_gradient_accum = torch.empty(0, 1, device=dev)
        self.denom = torch.empty(0, 1, device=dev)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.generation = 0        # bumped whenever the parameter tensors are replaced (prune / densify / restore)

    # ---- activations --------------------------------------------------------------------------------------------
    scaling_activation = staticmethod(torch.exp)
    scaling_inverse_activation = staticmethod(torch.log)
    opacity_activation = staticmethod(torch.sigmoid)
    inverse_opacity_activation = staticmethod(inverse_sigmoid)
    rotation_activation = staticmethod(torch.nn.functional.normalize)

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_rgb(self):
        return self._rgb

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        L = build_scaling_rotation(scaling_modifier * self.get_scaling, self._rotation)
        return strip_symmetric(L @ L.transpose(1, 2))

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- optimiser ----------------------------------------------------------------------------------------------
    def _params(self):
        return dict(xyz=self._xyz, f_dc=self._features_dc, f_rest=self._features_rest, opacity=self._opacity,
                    scaling=self._scaling, rotation=self._rotation, rgb=self._rgb)

    def _assign(self, t):
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation, self._rgb = t["opacity"], t["scaling"], t["rotation"], t["rgb"]

    def training_setup(self):
        m = self.cfg["mapping"]
        self.percent_dense = m["percent_dense"]
        self.spatial_lr_scale = m["spatial_lr_scale"]
        n, dev = self._xyz.shape[0], self.cfg["device"]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        lrs = dict(xyz=m["position_lr_init"] * self.spatial_lr_scale, f_dc=m["feature_lr"], f_rest=m["feature_lr"] / 20.0,
                   opacity=m["opacity_lr"], scaling=m["scaling_lr"], rotation=m["rotation_lr"], rgb=m["rgb_lr"])
        params = {k: nn.Parameter(v.detach().clone().requires_grad_(True)) for k, v in self._params().items()}
        self._assign(params)
        self.optimizer = torch.optim.Adam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in _GROUPS],
                                          lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=m["position_lr_init"] * self.spatial_lr_scale,
                                                    lr_final=m["position_lr_final"] * self.spatial_lr_scale,
                                                    lr_delay_mult=m["position_lr_delay_mult"],
                                                    max_steps=m["position_lr_max_steps"])

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
                return group["lr"]

    def _rebuild_groups(self, transform_param, transform_state):
        """Replace every group's parameter by ``transform_param(old)`` (a new leaf) and carry the Adam moments
        through ``transform_state`` -- the reference's cat/prune surgery."""
        out = {}
        self.generation += 1
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(transform_param(group["name"], old.detach()).requires_grad_(True))
            if state is not None and "exp_avg" in state:
                state["exp_avg"] = transform_state(group["name"], state["exp_avg"])
                state["exp_avg_sq"] = transform_state(group["name"], state["exp_avg_sq"])
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
        return out

    # ---- snapshot / restore (overflow recovery of the native mapping loop, fused.py) -----------------------------------
    def snapshot(self):
        """Copies of everything a mapping loop mutates: parameters, Adam moments and step counters, learning rates, the
        densification statistics.  ~24 device copies (26 MB at 150 k Gaussians)."""
        groups = []
        for group in self.optimizer.param_groups:
            p = group["params"][0]
            st = self.optimizer.state.get(p, {})
            groups.append((group["name"], p.detach().clone(), {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}, group["lr"]))
        return dict(groups=groups, stats=(self.xyz_gradient_accum.clone(), self.denom.clone(), self.max_radii2D.clone()))

    def restore(self, snap):
        """Back to a ``snapshot()`` (which stays valid: tensors are copied again)."""
        by_name = {name: (p, st, lr) for name, p, st, lr in snap["groups"]}
        out = {}
        self.generation += 1
        for group in self.optimizer.param_groups:
            self.optimizer.state.pop(group["params"][0], None)
            p, st, lr = by_name[group["name"]]
            new = nn.Parameter(p.clone().requires_grad_(True))
            if st:
                self.optimizer.state[new] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
            group["params"][0] = new
            group["lr"] = lr
            out[group["name"]] = new
        self._assign(out)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = (t.clone() for t in snap["stats"])

    # ---- device-side surgery (csrc/compact.hip through the C ABI; CUDA tensors only) ---------------------------------------
    def _native(self):
        return str(self.cfg["device"]).startswith("cuda")

    def prune_mask_device(self, min_opacity, extent, max_screen_size=None, counter_ptr=None):
        """The pruning predicate of ``prune`` evaluated by one kernel: returns the uint8 keep mask; the number of pruned
        Gaussians is ADDED to the device word at ``counter_ptr`` (no host synchronisation)."""
        import ctypes as C
        from . import _lib
        from .rasterizer import _stream
        lib = _lib.load()
        P = int(self._xyz.shape[0])
        keep = torch.empty(P, dtype=torch.uint8, device=self._xyz.device)
        if counter_ptr is None:
            self._prune_counter = torch.zeros(1, dtype=torch.int32, device=self._xyz.device)
            counter_ptr = self._prune_counter.data_ptr()
        radii = self.max_radii2D if max_screen_size is not None else None
        _lib.check(lib.mm3dgs_prune_mask(P, C.c_void_p(self._opacity.data_ptr()), C.c_void_p(self._scaling.data_ptr()),
                                         C.c_void_p(radii.data_ptr()) if radii is not None else None, float(min_opacity), float(0.1 * extent),
                                         float(max_screen_size if max_screen_size is not None else 0.0), C.c_void_p(keep.data_ptr()),
                                         C.c_void_p(counter_ptr), _stream()))
        return keep

    def compact_device(self, keep):
        """``prune_points(~keep)`` with the order-preserving compaction kernels: one plan (counts + scan), ONE 4-byte read-back
        for the new size, one scatter launch over every parameter, Adam moment and statistic (instead of a nonzero and ~24
        index_select launches).  Nothing to prune leaves every tensor in place, like ``prune_points``."""
        import ctypes as C
        from . import _lib
        from .rasterizer import _stream
        lib = _lib.load()
        P = int(keep.shape[0])
        dev = keep.device
        work = torch.empty(lib.mm3dgs_compact_work_bytes(P), dtype=torch.uint8, device=dev)
        n_keep = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mm3dgs_compact_plan(P, C.c_void_p(keep.data_ptr()), C.c_void_p(work.data_ptr()), C.c_void_p(n_keep.data_ptr()), _stream()))
        n = int(n_keep.item())
        if n == P:
            for group in self.optimizer.param_groups:
                group["params"][0].grad = None
            return
        pairs = []                      # (old tensor, new tensor)

        def moved(t):
            new = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            pairs.append((t.detach(), new))
            return new
        out = {}
        self.generation += 1
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(moved(old).requires_grad_(True))
            if state is not None and "exp_avg" in state:
                state["exp_avg"], state["exp_avg_sq"] = moved(state["exp_avg"]), moved(state["exp_avg_sq"])
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
        self._assign(out)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = moved(self.xyz_gradient_accum), moved(self.denom), moved(self.max_radii2D)
        table = (_lib.Mm3dgsCompactArray * 32)()
        k = 0
        for old, new in pairs:
            w = int(old[0].numel()) if old.shape[0] > 0 else 0
            if w == 0:
                continue
            table[k].src, table[k].dst, table[k].width = old.data_ptr(), new.data_ptr(), w
            k += 1
        if n == 0:
            return      # every Gaussian pruned: the empty tensors are in place (their data_ptr() is NULL, nothing to move)
        _lib.check(lib.mm3dgs_compact_rows(P, C.c_void_p(keep.data_ptr()), C.c_void_p(work.data_ptr()), table, k, _stream()))
        self._surgery_keepalive = (pairs, keep, work)      # the launch is asynchronous

    def seed_device(self, color, depth, mask, pose, fx, fy, cx, cy):
        """``densification_postfix`` with the rows of the new Gaussians written by the seeding kernel: one Gaussian per pixel of
        ``mask`` (bool [H,W], raster order), initialised as slam/mapper.py:437-474,644-668 do.  Returns the number added."""
        import ctypes as C
        from . import _lib
        from .rasterizer import _stream
        lib = _lib.load()
        dev = depth.device
        H, W = depth.shape
        keep = mask.reshape(-1).to(torch.uint8).contiguous()
        work = torch.empty(lib.mm3dgs_compact_work_bytes(H * W), dtype=torch.uint8, device=dev)
        n_new = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mm3dgs_compact_plan(H * W, C.c_void_p(keep.data_ptr()), C.c_void_p(work.data_ptr()), C.c_void_p(n_new.data_ptr()), _stream()))
        n = int(n_new.item())           # (a keyframe event: the new size must be known to allocate)
        P = int(self._xyz.shape[0])

        def grown(t, zero_tail):
            new = (torch.zeros if zero_tail else torch.empty)((P + n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            if P:
                new[:P].copy_(t.detach())
            return new
        out = {}
        self.generation += 1
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(grown(old, False).requires_grad_(True))
            if state is not None and "exp_avg" in state:
                state["exp_avg"], state["exp_avg_sq"] = grown(state["exp_avg"], True), grown(state["exp_avg_sq"], True)
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
        self._assign(out)
        so = _lib.Mm3dgsSeedOutputs()
        so.xyz, so.f_dc, so.f_rest = self._xyz.data_ptr(), self._features_dc.data_ptr(), self._features_rest.data_ptr()
        so.opacity, so.scaling, so.rotation, so.rgb = self._opacity.data_ptr(), self._scaling.data_ptr(), self._rotation.data_ptr(), self._rgb.data_ptr()
        color, depth, pose = color.contiguous().float(), depth.contiguous().float(), pose.detach().contiguous().float()
        if n:
            _lib.check(lib.mm3dgs_seed_gaussians(H, W, C.c_void_p(color.data_ptr()), C.c_void_p(depth.data_ptr()), C.c_void_p(keep.data_ptr()),
                                                 C.c_void_p(work.data_ptr()), C.c_void_p(pose.data_ptr()), float(fx), float(fy), float(cx), float(cy),
                                                 P, C.byref(so), int(self._features_rest.shape[1]), _stream()))
        self.xyz_gradient_accum = torch.zeros((P + n, 1), device=dev)
        self.denom = torch.zeros((P + n, 1), device=dev)
        self.max_radii2D = torch.zeros((P + n,), device=dev)
        self._surgery_keepalive = (keep, work, color, depth, pose)
        return n

    def prune_points(self, mask):
        # ONE nonzero (one host sync) shared by the ~24 tensors instead of a boolean-mask gather (= nonzero + sync) per
        # tensor; nothing to prune (the common case in SLAM: two pruning steps per frame) leaves every tensor, parameter
        # object and Adam moment in place -- the same values the rebuild would produce
        idx = torch.nonzero(~mask, as_tuple=False).squeeze(1)
        if idx.numel() == mask.numel():
            # the rebuild would hand the optimiser fresh parameters WITHOUT gradients, which is what makes the reference's
            # optimizer.step() of a pruning iteration a no-op: keep that
            for group in self.optimizer.param_groups:
                group["params"][0].grad = None
            return
        self._assign(self._rebuild_groups(lambda n, p: p.index_select(0, idx), lambda n, s: s.index_select(0, idx)))
        self.xyz_gradient_accum = self.xyz_gradient_accum.index_select(0, idx)
        self.denom = self.denom.index_select(0, idx)
        self.max_radii2D = self.max_radii2D.index_select(0, idx)

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation, new_rgb):
        ext = dict(xyz=new_xyz, f_dc=new_features_dc, f_rest=new_features_rest, opacity=new_opacities,
                   scaling=new_scaling, rotation=new_rotation, rgb=new_rgb)
        self._assign(self._rebuild_groups(lambda n, p: torch.cat((p, ext[n].to(p)), 0),
                                          lambda n, s: torch.cat((s, torch.zeros_like(ext[n]).to(s)), 0)))
        n, dev = self._xyz.shape[0], self.cfg["device"]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)

    def prune(self, min_opacity, extent, max_screen_size=None, lazy_mask=False):
        """Returns the mask of the pruned Gaussians; lazy_mask: a zero-argument callable that produces it (the native loop only needs it
        under bundle adjustment, and an operator launched after the read-back delays the next run by its dispatch time)."""
        if self._native():
            # predicate, plan and compaction on the device (three small launches + a 4-byte read-back of the new size)
            keep = self.prune_mask_device(min_opacity, extent, max_screen_size)
            self.compact_device(keep)
            return (lambda: keep == 0) if lazy_mask else keep == 0
        mask = (self.get_opacity < min_opacity).squeeze(-1)
        big = self.get_scaling.max(dim=1).values > 0.1 * extent
        if max_screen_size is not None:
            big = torch.logical_or(big, self.max_radii2D > max_screen_size)
        mask = torch.logical_or(mask, big)
        self.prune_points(mask)
        return (lambda: mask) if lazy_mask else mask

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1,
                                                             keepdim=True)
        self.denom[update_filter] += 1

    # ---- PLY (reference attribute order, slam/gaussian_model.py:205-257) ------------------------------------------
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
        names += [f"rgb_{i}" for i in range(3)]
        return names

    def save_ply(self, path):
        f = lambda t: t.detach().float().cpu().numpy()
        xyz = f(self._xyz)
        cols = [xyz, np.zeros_like(xyz), f(self._features_dc.transpose(1, 2).flatten(1)),
                f(self._features_rest.transpose(1, 2).flatten(1)), f(self._opacity), f(self._scaling), f(self._rotation),
                f(self._rgb)]
        data = np.concatenate(cols, axis=1).astype("<f4")
        names = self.construct_list_of_attributes()
        assert data.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {data.shape[0]}\n"
        header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
        with open(path, "wb") as fh:
            fh.write(header.encode("ascii"))
            fh.write(data.tobytes())

    def load_ply(self, path):
        with open(path, "rb") as fh:
            names, count = [], 0
            while True:
                line = fh.readline().decode("ascii").strip()
                if line.startswith("element vertex"):
                    count = int(line.split()[-1])
                elif line.startswith("property float"):
                    names.append(line.split()[-1])
                elif line == "end_header":
                    break
            data = np.frombuffer(fh.read(count * len(names) * 4), dtype="<f4").reshape(count, len(names))
        col = {n: i for i, n in enumerate(names)}
        dev = self.cfg["device"]
        grab = lambda prefix: torch.tensor(data[:, [col[n] for n in names if n.startswith(prefix)]].copy(), device=dev)
        n_rest = (self.max_sh_degree + 1) ** 2 - 1
        self._xyz = torch.tensor(data[:, [col["x"], col["y"], col["z"]]].copy(), device=dev)
        self._features_dc = grab("f_dc_").reshape(count, 3, 1).transpose(1, 2).contiguous()
        self._features_rest = grab("f_rest_").reshape(count, 3, n_rest).transpose(1, 2).contiguous()
        self._opacity = torch.tensor(data[:, [col["opacity"]]].copy(), device=dev)
        self._scaling = grab("scale_")
        self._rotation = grab("rot_")
        self._rgb = grab("rgb_") if any(n.startswith("rgb_") for n in names) else torch.zeros(count, 3, device=dev)
        self.max_radii2D = torch.zeros((count,), device=dev)
        self.active_sh_degree = self.max_sh_degree
