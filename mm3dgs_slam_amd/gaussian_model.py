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

    def prune(self, min_opacity, extent, max_screen_size=None):
        mask = (self.get_opacity < min_opacity).squeeze(-1)
        big = self.get_scaling.max(dim=1).values > 0.1 * extent
        if max_screen_size is not None:
            big = torch.logical_or(big, self.max_radii2D > max_screen_size)
        mask = torch.logical_or(mask, big)
        self.prune_points(mask)
        return mask

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
