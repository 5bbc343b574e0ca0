"""Native (HIP) tracking / mapping iterations: the reference's per-iteration torch graph -- pose algebra, means
transform, activations, two rasterizer passes, loss, backward, Adam (slam/tracker.py:94-177, slam/mapper.py:798-948) --
collapsed into a handful of C-ABI calls with no host synchronisation inside the loop:

    mm3dgs_slam_forward  ->  mm3dgs_loss  ->  mm3dgs_slam_backward ( -> pose Adam on device | mm3dgs_adam )

``FusedTracker`` / ``FusedMapper`` subclass the torch-graph ``Tracker`` / ``Mapper`` and take over ``optimize_cam`` /
``optimize_map`` whenever the covariances come from scales + rotations and the ACTIVE SH degree is 0 (the reference never
raises it): both branches of ``transform_means_python`` (round 4: world-frame means natively), models that carry SH rows,
``convert_SHs_python``, the IMU residual, bundle adjustment -- also with a sharded mapping window (round 4) -- and the
``method: splatam`` losses and pruning schedule, and (round 5) ``keep_best_candidate`` (the arg-min over the iterations' losses is
kept on the device).  What is left (``compute_cov3D_python``, a resumed checkpoint with an active SH degree > 0) falls back to the
torch-graph loop with a warning; that loop stays the parity reference for these kernels (tests/test_gpu_fused.py).
"""
from __future__ import annotations

import ctypes as C
import math
import time
from random import randint

import torch

from . import _lib
from .mapper import Mapper
from .rasterizer import _camera, _stream
from .tracker import Tracker


def _gauss_window():
    g = torch.tensor([math.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)])
    return (g / g.sum()).float().tolist()


_WINDOW = _gauss_window()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FusedEngine:
    """Owns the reusable device buffers of the fused render (state, output, gradients) for one Renderer."""
    MIN_PAIRS = 65536      # floor of the binning capacity (pairs)
    # longest tile list (seen so far) up to which the engine asks for the one-launch sort + compositing kernels and direct bins: they sort lists of up to
    # 2048 splats in LDS (1024: run sort + rank merge; 2048: bitonic), anything longer through global memory -- correct, but a tile that does holds its
    # whole launch up, and from there on the three-launch path with its 16 K-entry LDS tier is the better one.  (1400 until round 6: a dense map -- the
    # UTMM-shaped configuration grown to 320 k Gaussians, 600 .. 840 pairs per tile on average -- crossed it and fell back to three launches per render.)
    FAST_PATH_MAX_LIST = 2048
    DIRECT_BINS = True     # size the binning state as tiles x (per-tile capacity) so that projection + binning are one launch

    def __init__(self, renderer):
        self.r = renderer
        self.dev = torch.device(renderer.cfg["device"])
        self.lib = _lib.load()
        self.H, self.W = renderer.image_height, renderer.image_width
        # persistent and zero-initialised: the library leaves the header / tile counters zero after every forward
        self.img_state = torch.zeros(self.lib.mm3dgs_image_bytes(self.H, self.W), dtype=torch.uint8, device=self.dev)
        self.max_tile_len = 1 << 30
        self.out = torch.empty((6, self.H, self.W), device=self.dev)
        self.dL = torch.empty((6, self.H, self.W), device=self.dev)
        self.loss = torch.zeros(4, device=self.dev)
        self.loss_work = torch.empty(self.lib.mm3dgs_loss_work_bytes(self.H, self.W), dtype=torch.uint8, device=self.dev)
        self.dpose = torch.zeros(7, device=self.dev)
        self.ratio = None            # pairs per Gaussian seen so far (binning capacity model)
        self.P = -1
        self.n_cap = 0
        rs_cls = renderer.settings_cls
        eye = torch.eye(4, device=self.dev)
        self.settings = rs_cls(image_height=self.H, image_width=self.W, tanfovx=renderer.tanfovx, tanfovy=renderer.tanfovy,
                               bg=renderer.background, scale_modifier=1.0, viewmatrix=eye, projmatrix=renderer.projection_matrix,
                               sh_degree=0, campos=torch.zeros(3, device=self.dev), prefiltered=False, debug=False)
        self._keep = (eye, self.settings.campos)
        self.cam = _camera(self.settings, renderer.background, eye, renderer.projection_matrix.contiguous(), self.settings.campos)
        self.isotropic = 1 if renderer.cfg["pipeline"]["force_isotropic"] else 0

    _warned = set()

    @staticmethod
    def eligible(cfg, gaussians):
        pipe = cfg["pipeline"]
        # SH: what matters is the ACTIVE degree.  The reference never raises it during SLAM (oneupSHdegree is not called on the path;
        # only a map loaded from a checkpoint starts at max_sh_degree), so a model with `mapping.sh_degree` > 0 still renders
        # SH_C0 f_dc + 0.5 -- in the kernel or, with `convert_SHs_python`, in Python (slam/renderer.py:179-193: at degree 0 the viewing
        # direction does not enter) -- while its f_rest rows ride along through seeding / pruning with zero gradients (torch's Adam
        # leaves a zero-gradient parameter with zero moments where it is).  The native loops do exactly that (round 4).
        # transform_means_python: false (world-frame means, pose gradient through the view matrix) runs natively too: Mm3dgsSlamInputs.world_means
        # (with the reference's literal depth bundle; this repository's optional `fix_depth_transpose` only exists in the torch-graph renderer)
        fixed_depth = (not pipe["transform_means_python"]) and pipe.get("fix_depth_transpose", False)
        # round 6 (ABI 209): an ACTIVE degree above 0 (a map resumed from a checkpoint, slam/gaussian_model.py:363) runs natively too -- the kernels evaluate
        # the SH colour at the normalised camera-space mean (what slam/renderer.py:179-193 hands the rasterizer in the shipped mode: means pre-transformed,
        # campos = 0), step f_rest as a sixth Adam group and carry the direction's share of the pose / mean gradients -- in the shipped mode with the
        # kernel's own SH evaluation; `convert_SHs_python` (directions taken from the WORLD means in Python) and the world-frame mode stay on the torch-graph loops
        sh_native = gaussians.active_sh_degree == 0 or (pipe["transform_means_python"] and not pipe["convert_SHs_python"] and gaussians.active_sh_degree <= 3
                                                        and int(gaussians._features_rest.shape[1]) <= 15)
        ok = (not pipe["compute_cov3D_python"] and not fixed_depth and sh_native and str(cfg["device"]).startswith("cuda"))
        if not ok and str(cfg["device"]).startswith("cuda"):
            # (the fallback is ~30x slower and used to be silent)
            why = ", ".join(w for w, bad in (("pipeline.compute_cov3D_python: true", pipe["compute_cov3D_python"]),
                                             ("pipeline.fix_depth_transpose", fixed_depth),
                                             (f"active SH degree {gaussians.active_sh_degree} > 0 with convert_SHs_python / world-frame means", not sh_native)) if bad)
            if why not in FusedEngine._warned:
                FusedEngine._warned.add(why)
                import warnings
                warnings.warn(f"mm3dgs: this configuration ({why}) is outside the native SLAM loops (covariances from scales + rotations; an active SH degree > 0 only in the shipped mode with the kernel's SH evaluation); "
                              "tracking and mapping run the torch-graph loops around the generic HIP rasterizer -- correct, but about 30x slower")
        return ok

    def _ensure(self, P, need_grads):
        if P != self.P:
            # the map grows by a few percent per keyframe: buffers are sized for 1.25 P and re-used until outgrown (a fresh
            # hipMalloc of the ~0.5 GB scratch at every keyframe cost ~10 ms of the frame)
            if P > getattr(self, "_cap_P", -1):
                self._cap_P = int(P * 1.25) + 1024
                u8 = dict(dtype=torch.uint8, device=self.dev)
                self.geom = torch.empty(self.lib.mm3dgs_geom_bytes(self._cap_P), **u8)
                self._radii_buf = torch.empty(self._cap_P, dtype=torch.int32, device=self.dev)
                self.n_cap = 0
            self.radii = self._radii_buf[:P]
            self.P = P
            self.grads = None
        want = int((self.ratio if self.ratio is not None else 24.0) * max(P, 1) * 2.0) + self.MIN_PAIRS
        self.direct = False
        # direct bins (MM3DGS_FWD_DIRECT_BINS): every tile owns n_cap / T pairs, sized from the longest list seen so far; the key's low
        # word holds the Gaussian id and the slot in the span, so the span is limited to 2^(32 - bits(P)) - 1 pairs (8191 up to 512 k
        # Gaussians, 4095 at 1 M)
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        per_tile = int(self.max_tile_len * 1.5) + 128
        slot_bits = min(13, 32 - max(int(P - 1).bit_length(), 1)) if P > 0 else 0
        if self.DIRECT_BINS and self.max_tile_len <= self.FAST_PATH_MAX_LIST and slot_bits >= 10 and int(per_tile * 1.25) + 1 <= (1 << slot_bits) - 1:
            want = max(want, T * per_tile)
            # ... and every projection workgroup (256 Gaussians) 16 * n_cap / workgroups gradient records
            want = max(want, ((P + 255) // 256) * (int(getattr(self, "max_group_records", 0) * 1.5) + 1024) // 16 + 1)
            self.direct = True
        if want > self.n_cap or (self.ratio is not None and self.n_cap > 4 * want):
            u8 = dict(dtype=torch.uint8, device=self.dev)
            self.n_cap = int(want * 1.25) if self.ratio is not None else want
            self.binning = torch.empty(self.lib.mm3dgs_binning_bytes(self.n_cap), **u8)
            self.scratch = torch.empty(self.lib.mm3dgs_backward_scratch_bytes(self._cap_P, self.n_cap), **u8)
        if need_grads and self.grads is None:
            # one flat buffer [xyz 3P | f_dc 3P | opacity P | scaling 3P | rotation 4P | accum P | denom P]: the multi-GPU
            # window all-reduces it in a single collective (window_parallel.py)
            # (+ 64: a sharded optimiser step reduce-scatters world x S >= 14 P elements of it, S = the 4-aligned shard size)
            self.flat = torch.zeros(16 * P + 64, device=self.dev)
            self.acc = torch.zeros(14 * P, device=self.dev)      # window-batch mode: sum of the local views' gradients
            o = [0, 3 * P, 6 * P, 7 * P, 10 * P, 14 * P, 15 * P, 16 * P]
            v = lambda i, shape: self.flat[o[i]:o[i + 1]].view(shape)
            self.grads = dict(xyz=v(0, (P, 3)), f_dc=v(1, (P, 1, 3)), opacity=v(2, (P, 1)), scaling=v(3, (P, 3)), rotation=v(4, (P, 4)))
            self.stat_delta = (torch.zeros(P, device=self.dev), v(5, (P, 1)), v(6, (P, 1)))   # max radii | accum | denom
            self._rest_rows = None

    def rest_grad(self, g):
        """Gradient buffer of the f_rest rows [P, n_rest, 3] for a model at an active SH degree > 0 (None at degree 0): what
        Mm3dgsSlamGrads.d_f_rest points at whenever the other gradient outputs are set."""
        if int(getattr(g, "active_sh_degree", 0)) <= 0:
            return None
        shape = tuple(g._features_rest.shape)
        if getattr(self, "_rest_rows", None) is None or tuple(self._rest_rows.shape) != shape:
            self._rest_rows = torch.zeros(shape, device=self.dev)
            if self.grads is not None:
                self.grads["f_rest"] = self._rest_rows
        return self._rest_rows

    def _flags(self):
        """STATE_CLEAN | SHORT_LISTS (hint from the last header check) | DIRECT_BINS (the capacity was sized per tile)."""
        return _lib.FWD_STATE_CLEAN | (_lib.FWD_SHORT_LISTS if self.max_tile_len <= self.FAST_PATH_MAX_LIST else 0) | (_lib.FWD_DIRECT_BINS if self.direct else 0)

    def inputs(self, pose, g):
        si = _lib.Mm3dgsSlamInputs()
        si.pose = pose.data_ptr()
        si.xyz, si.f_dc, si.opacity = g._xyz.data_ptr(), g._features_dc.data_ptr(), g._opacity.data_ptr()
        si.scaling, si.rotation = g._scaling.data_ptr(), g._rotation.data_ptr()
        si.isotropic = self.isotropic
        si.world_means = 0 if self.r.cfg["pipeline"]["transform_means_python"] else 1
        deg = int(getattr(g, "active_sh_degree", 0))
        if deg > 0:      # (ABI 209) the rows the kernels evaluate / step natively
            si.f_rest, si.sh_degree, si.n_rest = g._features_rest.data_ptr(), deg, int(g._features_rest.shape[1])
        return si

    def forward(self, pose, g, need_grads=False):
        P = int(g._xyz.shape[0])
        self._ensure(P, need_grads)
        si = self.inputs(pose, g)
        self._last_g = g
        flags = self._flags()
        _lib.check(self.lib.mm3dgs_slam_forward(C.byref(self.cam), P, C.byref(si), _p(self.out), _p(self.radii), _p(self.geom),
                                                _p(self.img_state), _p(self.binning), self.n_cap, flags, _stream()))
        return si

    def track_loop(self, n_iter, pose, g, lcfg, gt_color, ref, pose_adam):
        """All tracking iterations of a frame enqueued by one C call (no Python in the loop)."""
        P = int(g._xyz.shape[0])
        self._ensure(P, False)
        si = self.inputs(pose, g)
        flags = self._flags()
        _lib.check(self.lib.mm3dgs_slam_track(n_iter, C.byref(self.cam), P, C.byref(si), _p(self.out), _p(self.radii), _p(self.geom),
                                              _p(self.img_state), _p(self.binning), self.n_cap, flags, C.byref(lcfg), _p(gt_color),
                                              _p(ref), _p(self.loss_work), _p(self.dL), _p(self.loss), _p(self.scratch),
                                              C.byref(pose_adam), _stream()))

    def adam_project(self, next_pose, g, grads, map_adam):
        """The multi-GPU window's optimiser step from the (all-reduced) gradient arrays and the projection + binning of the NEXT view in one
        launch (mm3dgs_slam_adam_project); the map_loop call that renders that view must say projected=True.  Direct bins only
        (self.can_adam_project())."""
        P = int(g._xyz.shape[0])
        si = self.inputs(next_pose, g)
        sg = _lib.Mm3dgsSlamGrads()
        sg.d_xyz, sg.d_f_dc, sg.d_opacity = grads["xyz"].data_ptr(), grads["f_dc"].data_ptr(), grads["opacity"].data_ptr()
        sg.d_scaling, sg.d_rotation = grads["scaling"].data_ptr(), grads["rotation"].data_ptr()
        self._pose_keepalive = next_pose
        _lib.check(self.lib.mm3dgs_slam_adam_project(C.byref(self.cam), P, C.byref(si), C.byref(sg), C.byref(map_adam), _p(self.radii), _p(self.geom),
                                                     _p(self.img_state), _p(self.binning), self.n_cap, self._flags(), _stream()))

    def can_adam_project(self, g):
        """Whether the library runs this map with direct bins under the flags the next calls will carry (the fused launch exists for
        them only; the library's own decision, not this class's sizing hint)."""
        P = int(g._xyz.shape[0])
        return P > 0 and bool(self.direct) and self.lib.mm3dgs_slam_direct_bins(C.byref(self.cam), P, self.n_cap, self._flags()) == 1

    def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, keep_tile_order=False, want_loss=True, projected=False):
        """A run of mapping iterations enqueued by one C call; views = [(pose[7], gt_color, ref_or_None), ...].  With `grads`
        (and map_adam None) the gradients of the last view are written out instead of stepped (multi-GPU window).
        keep_tile_order / want_loss=False: for callers that enqueue one iteration per call -- the workgroup -> tile table of an earlier
        call stays in force (MM3DGS_FWD_KEEP_TILE_ORDER) and the loss scalars' finishing launch is left out (`loss` then keeps the values
        of the last call that asked for them).  projected: adam_project already launched the projection + binning of views[0]."""
        P = int(g._xyz.shape[0])
        self._ensure(P, True)
        arr = getattr(views, "table", None)      # built ahead of time by FusedMapper (a _Views list)
        if arr is None:
            arr = self.view_table(views)
        si = self.inputs(views[0][0], g)
        sg = None
        if stats is not None or grads is not None:
            sg = _lib.Mm3dgsSlamGrads()
            if stats is not None:
                sg.max_radii2D, sg.grad_accum, sg.denom = (t.data_ptr() for t in stats)
            if grads is not None:
                sg.d_xyz, sg.d_f_dc, sg.d_opacity = grads["xyz"].data_ptr(), grads["f_dc"].data_ptr(), grads["opacity"].data_ptr()
                sg.d_scaling, sg.d_rotation = grads["scaling"].data_ptr(), grads["rotation"].data_ptr()
                rest = self.rest_grad(g)
                if rest is not None:
                    sg.d_f_rest = rest.data_ptr()
        flags = self._flags() | (_lib.FWD_KEEP_TILE_ORDER if keep_tile_order else 0) | (_lib.FWD_PROJECTED if projected else 0)
        self._views_keepalive = views      # the device work is asynchronous
        _lib.check(self.lib.mm3dgs_slam_map(len(views), arr, C.byref(self.cam), P, C.byref(si), _p(self.out), _p(self.radii), _p(self.geom),
                                            _p(self.img_state), _p(self.binning), self.n_cap, flags, C.byref(lcfg), _p(self.loss_work),
                                            _p(self.dL), _p(self.loss) if want_loss else None, _p(self.scratch), C.byref(sg) if sg is not None else None,
                                            C.byref(map_adam) if map_adam is not None else None, _stream()))

    @staticmethod
    def view_table(views):
        """The Mm3dgsMapView array of a run (pure host work: ~1 us per view).  FusedMapper builds the table of the run that FOLLOWS a
        pruning step before it blocks on that step's read-back, so the run is enqueued the moment the new map size is known."""
        arr = (_lib.Mm3dgsMapView * len(views))()
        for i, view in enumerate(views):
            pose, gt_color, ref = view[:3]
            arr[i].pose, arr[i].gt_color, arr[i].ref_depth_or_null = pose.data_ptr(), gt_color.data_ptr(), (ref.data_ptr() if ref is not None else None)
            if len(view) > 3 and view[3] is not None:        # bundle adjustment: this view's pose takes an Adam step on the device
                arr[i].pose_adam_or_null = C.addressof(view[3])
            if len(view) > 4 and view[4] is not None:        # ... or (sharded window) hands its pose gradient out for the reduce
                arr[i].dpose_out_or_null = view[4].data_ptr()
        return arr

    def visibility(self, pose, g, seen):
        """seen[i] += 1 for every Gaussian the projection stage would hand to the rasterizer from `pose` (radii > 0): the
        preprocess kernel alone, no binning / compositing (slam/mapper.py:690-716 needs only this)."""
        P = int(g._xyz.shape[0])
        self._ensure(P, False)
        si = self.inputs(pose, g)
        _lib.check(self.lib.mm3dgs_slam_visibility(C.byref(self.cam), P, C.byref(si), _p(self.radii), _p(seen), _p(self.geom), _stream()))

    def check_capacity(self):
        """Synchronises: reads the image-state header.  Its overflow / max_tile_len / max_num_rendered words are STICKY on the
        device (binning.hip: set by any forward since they were last cleared), so one read after a whole optimisation loop
        covers every iteration and every view of it.  Updates the capacity model from the LARGEST pair count seen, clears the
        sticky words, and returns False if any forward overflowed (its tile lists were clamped, so the loop's results are
        invalid: the caller restores its state and re-runs with the capacity this call has already raised)."""
        token = self.check_capacity_begin()
        torch.cuda.current_stream(self.dev).synchronize()
        return self.check_capacity_end(token)

    def check_capacity_begin(self):
        """First half of check_capacity, without the synchronisation: the header is copied to pinned host memory and its sticky words
        are cleared, in stream order.  A caller that is about to drain the stream anyway (the pruning step's 4-byte read-back) calls
        this before and check_capacity_end after: one round trip instead of two, and nothing launched between the read-back and the
        next run."""
        if getattr(self, "_hdr_pin", None) is None:
            self._hdr_pin = torch.empty(9, dtype=torch.int32).pin_memory()
        self._hdr_pin.copy_(self.img_state[:36].view(torch.int32), non_blocking=True)
        self.img_state[4:16].zero_()
        self.img_state[32:36].zero_()
        return (self.P, self.n_cap)

    def check_capacity_end(self, token):
        """Second half (host only): the stream must have been synchronised since check_capacity_begin."""
        h = self._hdr_pin
        P, n_cap = token
        overflow, n_max = int(h[1]), int(h[3])
        unchecked, self.unchecked_tracking = getattr(self, "unchecked_tracking", None), None
        if overflow and unchecked is not None:
            # a tracking loop that skipped its own header read (lazy check, ample headroom) ran -- at least in part -- on an overflowed
            # forward: the kernels voided those pose steps (no optimiser step, no corruption), but the frame was tracked with fewer
            # iterations than configured.  By now the map has moved on, so the loop cannot be re-run; counted and said aloud.
            import warnings
            self.unrecovered_tracking_overflows = getattr(self, "unrecovered_tracking_overflows", 0) + 1
            warnings.warn(f"mm3dgs: the tracking loop of frame {unchecked} ran without a capacity check and the binning capacity overflowed since "
                          "(its pose steps from the first overflowing forward on were skipped on the device); capacity raised")
        self.max_tile_len = int(h[2])
        self.max_group_records = max(getattr(self, "max_group_records", 0), int(h[8]))
        self.ratio = max(self.ratio or 0.0, n_max / max(P, 1))
        self.overflows = getattr(self, "overflows", 0) + (1 if overflow else 0)
        self._checked_P, self._checked_cap = P, n_cap
        return not overflow

    def headroom(self):
        """Smallest ratio capacity / (largest demand seen) over the three capacities a forward can run out of (pairs, per-tile
        span, gradient records per projection workgroup), for the CURRENT buffers and map size; 0 when nothing has been measured
        for them yet (fresh buffers, a map that changed size since the last check)."""
        cp = getattr(self, "_checked_P", -1)
        # (a map that shrank a little since the last check -- a pruning step -- can only need less)
        if self.ratio is None or self.P <= 0 or not (0.9 * cp <= self.P <= cp) or getattr(self, "_checked_cap", -1) != self.n_cap:
            return 0.0
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        h = self.n_cap / max(self.ratio * self.P, 1.0)
        if self.direct:
            h = min(h, (self.n_cap // T) / max(self.max_tile_len, 1))
            h = min(h, (16 * self.n_cap // max((self.P + 255) // 256, 1)) / max(getattr(self, "max_group_records", 0), 1))
        return h

    def loss_call(self, cfg, gt_color, ref):
        _lib.check(self.lib.mm3dgs_loss(C.byref(cfg), _p(self.out), _p(gt_color), _p(ref), _p(self.loss_work), _p(self.dL),
                                        _p(self.loss), _stream()))

    def backward(self, si, grads=None, stats=None, dpose=None, pose_adam=None, map_adam=None):
        sg = _lib.Mm3dgsSlamGrads()
        if grads is not None:
            sg.d_xyz, sg.d_f_dc, sg.d_opacity = grads["xyz"].data_ptr(), grads["f_dc"].data_ptr(), grads["opacity"].data_ptr()
            sg.d_scaling, sg.d_rotation = grads["scaling"].data_ptr(), grads["rotation"].data_ptr()
            rest = self.rest_grad(self._last_g) if getattr(self, "_last_g", None) is not None else None
            if rest is not None:
                sg.d_f_rest = rest.data_ptr()
        if stats is not None:
            sg.max_radii2D, sg.grad_accum, sg.denom = (t.data_ptr() for t in stats)
        _lib.check(self.lib.mm3dgs_slam_backward(C.byref(self.cam), self.P, C.byref(si), _p(self.radii), _p(self.geom), _p(self.img_state),
                                                 _p(self.binning), self.n_cap, _p(self.dL), _p(self.scratch), C.byref(sg), _p(dpose),
                                                 C.byref(pose_adam) if pose_adam is not None else None,
                                                 C.byref(map_adam) if map_adam is not None else None, self._flags(), _stream()))


class _Views(list):
    """The views of a run with their Mm3dgsMapView table (FusedEngine.view_table) attached: a plain list for every other consumer."""
    table = None


def _loss_cfg(H, W, w_l1, w_ssim, w_pearson, l1_mask, pearson_mask, invert, sil_thr, w_depth_l1=0.0, depth_l1_mask=0, l1_sum=0):
    c = _lib.Mm3dgsLossConfig()
    c.H, c.W, c.w_l1, c.w_ssim, c.w_pearson = H, W, w_l1, w_ssim, w_pearson
    c.l1_mask, c.pearson_mask, c.pearson_invert, c.sil_thr = l1_mask, pearson_mask, invert, sil_thr
    c.w_depth_l1, c.depth_l1_mask, c.l1_sum = w_depth_l1, depth_l1_mask, l1_sum      # the `method: splatam` forms (include/mm3dgs.h)
    for i, v in enumerate(_WINDOW):
        c.window[i] = v
    return c


def _engine(renderer):
    eng = getattr(renderer, "_fused_engine", None)
    if eng is None:
        eng = renderer._fused_engine = FusedEngine(renderer)
    return eng


class FusedTracker(Tracker):
    lazy_checks = True       # False: read the capacity header back after every loop (debugging / tests)

    def optimize_cam(self, idx, num_iter, optimizer, camera_tensor_q, camera_tensor_T, gt_color, gt_depth=None, est_depth=None):
        trk = self.cfg["tracking"]
        # (keep_best_candidate -- this repository's option for what slam/tracker.py:88-91,161-181 computes and then discards -- runs natively
        #  since round 5: the pose-finish kernel keeps the arg-min over the iterations' losses on the device, Mm3dgsPoseAdam.best)
        if (num_iter == 0 or not FusedEngine.eligible(self.cfg, self.gaussians)):
            return super().optimize_cam(idx, num_iter, optimizer, camera_tensor_q, camera_tensor_T, gt_color, gt_depth, est_depth)
        eng = _engine(self.renderer)
        dev = eng.dev
        t_start = time.perf_counter()
        with torch.no_grad():
            pose0 = torch.cat([camera_tensor_q.detach(), camera_tensor_T.detach()]).float().contiguous().clone()
            w_p, pmask, ref = 0.0, 0, None
            if trk["use_depth_estimate_loss"]:
                w_p = float(trk["pearson_weight"])
                pmask, ref = (1, est_depth) if not self.cfg["use_gt_depth"] else (3, gt_depth)
            lcfg = _loss_cfg(eng.H, eng.W, 1.0, 0.0, w_p, 1, pmask, 1, 0.99)
            if self.cfg["method"].lower() == "splatam":
                # slam/tracker.py:110-126: sum |gt_depth - depth| + 0.5 sum |gt - image| over { gt_depth > 0, silhouette > 0.99 }
                lcfg = _loss_cfg(eng.H, eng.W, 0.5, 0.0, 0.0, 3, 0, 0, 0.99, w_depth_l1=1.0, depth_l1_mask=3, l1_sum=1)
                ref = gt_depth
            gt_color = gt_color.contiguous()
            ref = None if ref is None else ref.contiguous()
            g = self.gaussians
            for attempt in range(4):
                # everything the loop mutates (pose, Adam moments, step) is rebuilt from the starting pose, so a binning
                # overflow anywhere in the loop (sticky header flag) is answered by re-running it with the raised capacity
                pose = pose0.clone()
                m, v = torch.zeros(7, device=dev), torch.zeros(7, device=dev)
                step = torch.zeros(1, dtype=torch.int32, device=dev)
                ad = _lib.Mm3dgsPoseAdam()
                ad.pose, ad.m, ad.v, ad.step = pose.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr()
                ad.lr_q, ad.lr_t = float(trk["rotation_lr"]), float(trk["position_lr"])
                ad.beta1, ad.beta2, ad.eps = 0.9, 0.999, 1e-8
                best = None
                if self.keep_best_candidate:
                    best = torch.cat([torch.full((1,), 1e20, device=dev), pose0]).contiguous()      # { loss, pose }: slam/tracker.py:88-91
                    ad.best = best.data_ptr()
                if trk["use_imu_loss"] and self.cfg["method"].lower() != "splatam":      # rel_pose_loss against the pose the optimisation starts from (slam/tracker.py:87,146-155; not in the splatam branch)
                    ad.prior_pose, ad.prior_w_t, ad.prior_w_q = pose0.data_ptr(), float(trk["imu_T_weight"]), float(trk["imu_q_weight"])
                eng.track_loop(num_iter, pose, g, lcfg, gt_color, ref, ad)
                # With ample headroom (largest demand seen < 2/3 of every capacity, same map size) the read-back -- a device
                # synchronisation per frame -- is left to the mapper's next drained point (the header's overflow word is sticky, so
                # nothing is lost; a frame's 100 pose steps cannot grow a tile list by half).
                if self.lazy_checks and getattr(eng, "headroom", lambda: 0.0)() >= 1.5:
                    # (should such a loop overflow after all, its pose steps from the first overflowing forward on are void
                    #  on the device -- the tracked pose is then the pose of the last complete iteration; the engine remembers that this
                    #  frame went unchecked and says so at the next header read, check_capacity_end)
                    eng.unchecked_tracking = idx
                    break
                if eng.check_capacity():
                    break
            else:
                raise RuntimeError("mm3dgs: tracking loop kept overflowing its binning capacity")
            if best is not None:
                pose = best[1:8]
            camera_tensor_q.data.copy_(pose[:4])
            camera_tensor_T.data.copy_(pose[4:])
            if self.cfg["debug"]["get_runtime_stats"]:
                # the reference times each iteration on the host (slam/tracker.py:100,164-168); the native loop has no host iterations, so
                # the whole loop is timed once (device-synchronised) and the averages time_sum / iter_count keep their meaning
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                self.tracking_time_sum += time.perf_counter() - t_start
            self.tracking_iter_count += num_iter
            return eng.loss[0].clone(), eng.out[:3].clone()


class FusedMapper(Mapper):
    fuse_adam_project = True     # multi-GPU window: the optimiser step and the next view's projection + binning in one launch (False: mm3dgs_adam, then the projection)
    lazy_checks = True       # False: read the capacity header back after every loop (debugging / tests)
    _piggybacked = False
    _late_overflow = False
    always_snapshot = False      # debugging / tests: snapshot the map before every mapping loop (the round-2 behaviour)

    def _render_depth_sil(self, pose):
        # the keyframe test renders once per frame: one native forward instead of the ~40 torch launches of Renderer.render
        if not FusedEngine.eligible(self.cfg, self.gaussians):
            return super()._render_depth_sil(pose)
        eng = _engine(self.renderer)
        if self._defer_render_check:
            # keyframe test: its two counters are read back a moment later (covisibility_ratio_dense); the capacity header of this render
            # rides on that read-back instead of draining the device a second time
            eng.forward(pose.detach().float().contiguous(), self.gaussians)
            self._pending_render_check = (eng.check_capacity_begin(), pose)
            return eng.out[3], eng.out[4]
        for _ in range(4):
            eng.forward(pose.detach().float().contiguous(), self.gaussians)
            if eng.check_capacity():       # (the keyframe test synchronises on its result anyway)
                return eng.out[3], eng.out[4]
        raise RuntimeError("mm3dgs: render kept overflowing its binning capacity")

    _defer_render_check = False
    _pending_render_check = None

    def need_new_keyframe(self, idx, est_pose, gt_color, gt_depth=None, est_depth=None):
        self._defer_render_check = hasattr(_engine(self.renderer), "check_capacity_begin") if FusedEngine.eligible(self.cfg, self.gaussians) else False
        try:
            return super().need_new_keyframe(idx, est_pose, gt_color, gt_depth, est_depth)
        finally:
            self._defer_render_check, self._pending_render_check = False, None

    def covisibility_ratio_dense(self, depth, sil, kf_pose, cur_pose):
        """The keyframe test's covisibility ratio (slam/mapper.py:141-216) as ONE kernel over the rendered depth / silhouette
        planes (mm3dgs_covisibility_ratio) instead of ~30 element-wise torch launches: two counters come back."""
        if not (FusedEngine.eligible(self.cfg, self.gaussians) and depth.is_cuda and depth.is_contiguous() and sil.is_contiguous()):
            return super().covisibility_ratio_dense(depth, sil, kf_pose, cur_pose)
        eng = _engine(self.renderer)
        fx, fy, cx, cy = self._intr()
        H, W = depth.shape
        counts = torch.empty(2, dtype=torch.int32, device=depth.device)
        kp, cp = kf_pose.detach().float().to(depth.device).contiguous(), cur_pose.detach().float().to(depth.device).contiguous()
        _lib.check(eng.lib.mm3dgs_covisibility_ratio(H, W, _p(depth), _p(sil), _p(kp), _p(cp), float(fx), float(fy), float(cx), float(cy),
                                                     _p(counts), _stream()))
        c = counts.cpu()                      # (the caller synchronises on the decision anyway)
        pending, self._pending_render_check = self._pending_render_check, None
        if pending is not None and not eng.check_capacity_end(pending[0]):
            # the render these planes came from overflowed its binning capacity (now raised): render again, checked, and test that
            self._defer_render_check = False
            depth, sil = self._render_depth_sil(pending[1])
            return self.covisibility_ratio_dense(depth, sil, kf_pose, cur_pose)
        return torch.tensor(float(c[0]) / max(float(c[1]), 1.0))

    def update_covisibility_graph(self, key):
        """A new keyframe's edges (slam/mapper.py:218-236 with get_depth_pointcloud :175-196 and is_covisible :198-216): its surface
        points -- one native render -- projected into every earlier keyframe by the covisibility kernel (one launch per keyframe, the
        counters of all of them read back at once).  The torch formulation this replaces rendered through the generic path and
        synchronised once per earlier keyframe (measured with a dozen keyframes: 13.6 ms per keyframe event)."""
        n_prev = len(self.keyframes) - 1
        if n_prev <= 0 or not FusedEngine.eligible(self.cfg, self.gaussians):
            return super().update_covisibility_graph(key)
        eng = _engine(self.renderer)
        if not hasattr(eng, "check_capacity_begin"):
            return super().update_covisibility_graph(key)
        import numpy as np
        fx, fy, cx, cy = self._intr()
        with torch.no_grad():
            kpose = self.keyframes[key].pose.detach().float().contiguous()
            others = [kf.pose.detach().float().to(eng.dev).contiguous() for kf in self.keyframes[:-1]]
            for _ in range(4):
                eng.forward(kpose, self.gaussians)
                token = eng.check_capacity_begin()
                counts = torch.empty(n_prev, 2, dtype=torch.int32, device=eng.dev)
                for kid, cp in enumerate(others):
                    _lib.check(eng.lib.mm3dgs_covisibility_ratio(eng.H, eng.W, _p(eng.out[3]), _p(eng.out[4]), _p(kpose), _p(cp), float(fx), float(fy),
                                                                 float(cx), float(cy), _p(counts[kid]), _stream()))
                c = counts.cpu().numpy()              # (the one synchronisation of the keyframe's graph update)
                if eng.check_capacity_end(token):
                    break
            else:
                raise RuntimeError("mm3dgs: render kept overflowing its binning capacity")
        thr = np.float32(self.cfg["mapping"]["kf_covisibility"])
        for kid in range(n_prev):
            # (float32 like the torch expression `inside.sum() / max(n, 1) > threshold`)
            if np.float32(c[kid, 0]) / np.float32(max(int(c[kid, 1]), 1)) > thr:
                self.covisibility_graph[key].add(kid)
                self.covisibility_graph[kid].add(key)

    def get_covisible_gaussians(self, keyframe_idx_list, curr_camera_tensor, min_kf=2):
        """Gaussians visible from >= 2 views of the window (slam/mapper.py:690-716; the reference ignores `min_kf` and uses 2):
        one projection-only launch per view accumulating a per-Gaussian counter -- no render."""
        if not FusedEngine.eligible(self.cfg, self.gaussians):
            return super().get_covisible_gaussians(keyframe_idx_list, curr_camera_tensor, min_kf)
        eng = _engine(self.renderer)
        g = self.gaussians
        with torch.no_grad():
            seen = torch.zeros(g._xyz.shape[0], dtype=torch.int32, device=eng.dev)
            keep = []
            for k in keyframe_idx_list:
                pose = (curr_camera_tensor if k == -1 else self.keyframes[k].pose).detach().float().contiguous()
                keep.append(pose)
                eng.visibility(pose, g, seen)
            self._vis_keepalive = keep
            return seen >= 2

    def initialize_new_gaussians(self, idx, camera_pose, gt_color, gt_depth=None, est_depth=None):
        """New-keyframe seeding (slam/mapper.py:409-493,600-688) with the non-presence test on one native render and the new rows
        written by the seeding kernel (csrc/compact.hip) instead of meshgrid / boolean-mask gathers / RGB2SH / cat."""
        if not FusedEngine.eligible(self.cfg, self.gaussians) or not self.gaussians._native():
            return super().initialize_new_gaussians(idx, camera_pose, gt_color, gt_depth, est_depth)
        depth = gt_depth if self.cfg["use_gt_depth"] else est_depth
        dev = depth.device
        with torch.no_grad():
            if idx == 0 and "iteration" not in self.cfg:
                non_presence = torch.ones(depth.numel(), dtype=torch.bool, device=dev)
            else:
                rdepth, sil = self._render_depth_sil(camera_pose)
                err = (depth - rdepth).abs() * (depth > 0)
                if self.cfg["method"].lower() == "splatam":     # slam/mapper.py:520-526: only surfaces IN FRONT of the map, 50 x median
                    far = (rdepth > depth) & (err > 50 * err.median())
                else:
                    far = err > 10 * err.median()
                non_presence = ((sil < 0.5) | far).reshape(-1)
            non_presence = non_presence & (depth > 0).reshape(-1)
            if self.cfg["method"].lower() == "splatam" and not bool(non_presence.any()):
                return None, non_presence.reshape(depth.shape)          # (slam/mapper.py:532,590-591)
            frac = float(self.cfg["mapping"].get("seed_fraction", 1.0))
            if frac < 1.0:     # workload knob (not in the reference): seed only a fixed pseudo-random subset of the pixels
                from .mapper import seed_subset
                non_presence = non_presence & seed_subset(non_presence.numel(), idx, frac, dev)
            fx, fy, cx, cy = self._intr()
            P0 = int(self.gaussians.get_xyz.shape[0])
            n = self.gaussians.seed_device(gt_color, depth, non_presence.reshape(depth.shape), camera_pose, fx, fy, cx, cy)
            new_mask = torch.zeros(P0 + n, dtype=torch.bool, device=dev)
            new_mask[P0:] = True
        return new_mask, non_presence.reshape(depth.shape)

    def optimize_map(self, idx, num_iter, keyframe_idx_list, new_gaussians_mask, curr_camera_tensor, curr_gt_color,
                     curr_gt_depth=None, curr_est_depth=None):
        m = self.cfg["mapping"]
        do_ba = bool(m["do_BA"]) and idx > 0
        sh_window = int(self.gaussians.active_sh_degree) > 0 and self.window is not None      # (the sharded window's flat gradient layout has no f_rest rows)
        if num_iter == 0 or sh_window or not FusedEngine.eligible(self.cfg, self.gaussians):
            return super().optimize_map(idx, num_iter, keyframe_idx_list, new_gaussians_mask, curr_camera_tensor, curr_gt_color,
                                        curr_gt_depth, curr_est_depth)
        eng = _engine(self.renderer)
        t_start = time.perf_counter()
        g = self.gaussians
        lam = float(m["lambda_dssim"])
        w_p, pmask = 0.0, 0
        if m["use_depth_estimate_loss"]:
            w_p = float(m["pearson_weight"])
            pmask = 0 if not self.cfg["use_gt_depth"] else 2
        lcfg = _loss_cfg(eng.H, eng.W, 1.0 - lam, lam, w_p, 0, pmask, 0, 0.5)
        splatam = self.cfg["method"].lower() == "splatam"
        if splatam:
            # slam/mapper.py:836-855: mean |gt_depth - depth| over { gt_depth > 0 } + 0.5 ((1-l) L1 + l (1 - SSIM)); no Pearson term
            w_p = 0.0
            lcfg = _loss_cfg(eng.H, eng.W, 0.5 * (1.0 - lam), 0.5 * lam, 0.0, 0, 0, 0, 0.5, w_depth_l1=1.0, depth_l1_mask=2)
        stack = None

        def pop():
            nonlocal stack
            if not stack:
                stack = list(keyframe_idx_list)
            return stack.pop(randint(0, len(stack) - 1))

        view_cache = {}

        def view_of(k):
            # the window holds a handful of distinct views; the ~150 picks of a frame reuse their (pose, target, reference)
            # tensors instead of re-deriving them with torch calls while the GPU waits for the next run to be enqueued
            if k not in view_cache:
                view_cache[k] = _view_of(k)
            return view_cache[k]

        ba_state = {}     # view id -> (pose buffer, m, v, step, Mm3dgsPoseAdam): bundle adjustment steps every window pose on the device

        def _view_of(k):
            if k == -1:
                pose, gt_color, gt_depth, est_depth = curr_camera_tensor, curr_gt_color, curr_gt_depth, curr_est_depth
            else:
                kf = self.keyframes[k]
                pose, gt_color, gt_depth, est_depth = kf.pose, kf.gt_color, kf.gt_depth, kf.est_depth
            ref = None
            if w_p:
                ref = (est_depth if not self.cfg["use_gt_depth"] else gt_depth).contiguous()
            if splatam:
                ref = gt_depth.contiguous()
            buf = pose.detach().float().contiguous()
            # Reference quirk kept by default: slam/mapper.py:752-760 puts `keyframes[k].pose[:4].requires_grad_()` views into the pose
            # optimiser, but the loop renders from FRESH views `keyframe.pose[:4]` (:817-819) that do not require grad -- so the
            # reference's bundle adjustment only ever refines the CURRENT frame's pose.  mapping.ba_optimize_keyframes: true gives
            # the intended behaviour (every window pose stepped when its view is rendered).
            if not do_ba or (k != -1 and not m.get("ba_optimize_keyframes", False)):
                return buf, gt_color.contiguous(), ref
            if multi:
                # sharded window (round 4): the view hands its pose gradient out; the gradients of a step's views are summed over the
                # ranks and every replica takes the identical pose step (_ba_window_step below) -- the state was created up front
                st = ba_state[k]
                return st[0], gt_color.contiguous(), ref, None, st[6]
            # pose Adam of slam/mapper.py:742-752: Adam(lr=0, eps=1e-15), groups cam_rot (cam_q_lr) / cam_pos (cam_t_lr); a pose is
            # only stepped in the iterations that render its view (torch skips parameters without a gradient), hence per-view state
            buf = buf.clone()
            mom, var = torch.zeros(7, device=eng.dev), torch.zeros(7, device=eng.dev)
            step = torch.zeros(1, dtype=torch.int32, device=eng.dev)
            ad = _lib.Mm3dgsPoseAdam()
            ad.pose, ad.m, ad.v, ad.step = buf.data_ptr(), mom.data_ptr(), var.data_ptr(), step.data_ptr()
            ad.lr_q, ad.lr_t, ad.beta1, ad.beta2, ad.eps = float(m["cam_q_lr"]), float(m["cam_t_lr"]), 0.9, 0.999, 1e-15
            ba_state[k] = (buf, mom, var, step, ad, pose)
            return buf, gt_color.contiguous(), ref, ad

        def prune_at(it):
            if splatam:      # "Splatam does not densify -- only prunes" (slam/mapper.py:879-884): iterations 0 and 20, no size threshold
                return it <= 20 and it % 20 == 0
            return it <= m["densify_until_iter"] and it >= m["densify_from_iter"] and it % m["pruning_interval"] == 0

        import random as _random
        self._opt_mask = None
        if do_ba:
            with torch.no_grad():
                om = self.get_covisible_gaussians(keyframe_idx_list, curr_camera_tensor, 2)
                if new_gaussians_mask is not None:
                    om = om | new_gaussians_mask
                self._opt_mask = om.to(torch.uint8).contiguous()
        multi = self.window is not None and self.window.sharded
        self._ba_ids = []
        if do_ba and multi:
            # bundle adjustment with a sharded window, natively: pose, moments, step counter and a gradient slot for EVERY optimisable
            # pose of the window, created on every rank in the same order (the views a rank renders differ, the state must not)
            def ba_init():
                ba_state.clear()
                ids = sorted(set(k for k in keyframe_idx_list if k == -1 or m.get("ba_optimize_keyframes", False)))
                self._ba_ids = ids
                self._ba_grad = torch.zeros(max(len(ids), 1), 7, device=eng.dev)
                for k in ids:
                    pose = curr_camera_tensor if k == -1 else self.keyframes[k].pose
                    buf = pose.detach().float().contiguous().clone()
                    ba_state[k] = (buf, torch.zeros(7, device=eng.dev), torch.zeros(7, device=eng.dev), [0], None, pose, torch.zeros(7, device=eng.dev))
            ba_init()
            self._ba_state, self._ba_init = ba_state, ba_init
        # overflow recovery: a forward whose (tile, splat) pairs exceed the binning capacity renders clamped lists (flagged
        # sticky in the header).  The loop below is read back once, at its end; if any of its forwards overflowed, the map,
        # the optimiser, the statistics and the keyframe-pick RNG are put back and the loop is re-run (capacity raised).
        # (Pruning steps are NOT speculated on: measured on the benchmark sequence, 28 % of the frames prune something, and a
        # re-run of the whole loop costs far more than the 4-byte read-back of the new size that an exact step needs.)
        # The snapshot (~24 device copies: every parameter, moment and statistic) is only taken when an overflow is conceivable: after
        # the map changed size (a keyframe seeded, a step pruned), on fresh buffers, or when the largest demand seen is within 1.5x
        # of a capacity (a frame's 150 Adam steps at the shipped learning rates cannot grow a tile list by that much).  Should a loop
        # overflow without one, the kernels void every iteration from the first overflowing forward on (the backward projection and
        # the pose step read the sticky header word and skip gradients, statistics and optimiser steps: csrc/fused.hip slam_bwd_body),
        # so the map stays as the last complete iteration left it -- the rest of the loop is lost, not corrupted: counted in
        # `unrecovered_overflows`, warned about, capacity raised for the next frame.
        eng._ensure(int(g._xyz.shape[0]), True)
        need_snap = self.always_snapshot or getattr(eng, "headroom", lambda: 0.0)() < 1.5 or (self.window is not None and self.window._collective)
        snap, rng_state = (g.snapshot(), _random.getstate()) if need_snap else (None, None)
        self._late_overflow, self._piggybacked = False, False
        for attempt in range(4):
            self._map_loop_once(eng, g, m, lcfg, num_iter, multi, pop, view_of, prune_at, piggyback=snap is None)
            if snap is None and self.lazy_checks and self._piggybacked and getattr(eng, "headroom", lambda: 0.0)() >= 1.5:
                # no snapshot was needed (ample headroom) and the header was read at this loop's pruning steps, where the device is
                # drained anyway: the end-of-loop read-back (one more synchronisation per frame) is left to the next drained point
                ok = not self._late_overflow
            else:
                ok = eng.check_capacity() and not self._late_overflow
            if self.window is not None and self.window._collective:
                ok = not self.window.any_flag(not ok, device=eng.dev)
            if ok:
                break
            if snap is None:
                import warnings
                self.unrecovered_overflows = getattr(self, "unrecovered_overflows", 0) + 1
                warnings.warn("mm3dgs: a mapping loop overflowed its binning capacity without a snapshot to restore (its iterations from the "
                              "first overflowing forward on were skipped on the device: no optimiser step taken); capacity raised")
                break
            self.loop_reruns = getattr(self, "loop_reruns", 0) + 1
            g.restore(snap)
            self._moments_stale = False      # (the snapshot was taken with whole replicas: sharded optimiser steps of the failed attempt are gone with it)
            _random.setstate(rng_state)
            stack = None
            view_cache.clear(); ba_state.clear()       # (pose buffers and their Adam state start over)
            if do_ba and multi:
                self._ba_init()
            if do_ba:
                with torch.no_grad():
                    om = self.get_covisible_gaussians(keyframe_idx_list, curr_camera_tensor, 2)
                    if new_gaussians_mask is not None:
                        om = om | new_gaussians_mask
                    self._opt_mask = om.to(torch.uint8).contiguous()
        else:
            raise RuntimeError("mm3dgs: mapping loop kept overflowing its binning capacity")
        if do_ba:      # the optimised window poses go back where the reference's in-place Adam leaves them
            with torch.no_grad():
                for k, st in ba_state.items():
                    st[5].data.copy_(st[0])
        if self.cfg["debug"]["get_runtime_stats"]:
            if eng.dev.type == "cuda":
                torch.cuda.synchronize(eng.dev)
            self.mapping_time_sum += time.perf_counter() - t_start
        self.mapping_iter_count += num_iter

    def _map_loop_once(self, eng, g, m, lcfg, num_iter, multi, pop, view_of, prune_at, piggyback=False):
        splatam = self.cfg["method"].lower() == "splatam"       # prunes (prune_at), never collects densification statistics

        def dens(it):
            return (not splatam) and it <= m["densify_until_iter"]

        def run_length(start):
            n = 1
            while start + n < num_iter and not prune_at(start + n) and dens(start + n) == dens(start):
                n += 1
            return n

        try:
            self._map_loop_body(eng, g, m, lcfg, num_iter, multi, pop, view_of, prune_at, piggyback, splatam, dens, run_length)
        except BaseException:
            #  an exception after sharded optimiser steps -- a failing check, a prune that raises, a collective error -- must not
            # leave every rank with fresh Adam moments for its own slice only: SLAM.run goes on to write the map.  Gather them if the
            # process group still answers; if it does not, say so in the state the writers look at.
            if multi and getattr(self, "_moments_stale", False):
                try:
                    self._sync_moments(eng)
                except Exception:      # noqa: BLE001 -- the original exception is the one to report
                    self.window_incoherent = True
            raise

    def _map_loop_body(self, eng, g, m, lcfg, num_iter, multi, pop, view_of, prune_at, piggyback, splatam, dens, run_length):
        with torch.no_grad():
            iteration = 0
            prepared = None      # (first iteration, views, table) of a run whose host side was built while the GPU was still busy
            pending, projected_next = None, False      # multi-GPU window: the next step's keyframe picks, drawn ahead by a fused Adam + projection launch
            while iteration < num_iter:
                densify = dens(iteration)
                if not multi and not prune_at(iteration):
                    # single GPU: the run of iterations up to the next pruning step (or the end of the densification phase)
                    # is enqueued by ONE C call -- no Python between the ~9 launches of an iteration
                    if prepared is not None and prepared[0] == iteration:
                        views = prepared[1]
                    else:
                        views = _Views(view_of(pop()) for _ in range(run_length(iteration)))
                        views.table = FusedEngine.view_table(views)
                    prepared = None
                    n = len(views)
                    stats = (g.max_radii2D, g.xyz_gradient_accum, g.denom) if densify else None
                    eng.map_loop(views, g, lcfg, stats, self._inline_adam(n))
                    iteration += n
                    continue
                # one optimiser step over this rank's share of the window batch (a single view without a window)
                if pending is not None:
                    all_ids, ids = pending        # (popped ahead by the previous step, whose Adam launch already projected ids[0]'s view)
                else:
                    all_ids, ids = self.window.take_all(pop) if self.window is not None else (None, [pop()])
                was_projected, pending, projected_next = projected_next, None, False
                P = int(g._xyz.shape[0])
                eng._ensure(P, True)
                prune_now = prune_at(iteration)
                if multi:
                    # forward, loss and backward of each view in one C call (gradients written out, not stepped); more than
                    # one local view (window-batch mode): summed in view order; then ONE flat all-reduce over the ranks carries
                    # every parameter gradient and, while densifying, the statistics tail [14P, 16P) (+ a max-reduce of the radii)
                    if densify:
                        torch._foreach_zero_([eng.stat_delta[0], eng.flat[14 * P:]])        # (one launch)
                    if self._ba_ids:
                        self._ba_grad.zero_()
                    for j, k in enumerate(ids):
                        # (one iteration per C call here: the workgroup -> tile table is rebuilt by the first call of the loop only, the
                        #  loss scalars are finished by the last one only -- 9 + 6 us of launches per step otherwise)
                        eng.map_loop([view_of(k)], g, lcfg, eng.stat_delta if densify else None, None, grads=eng.grads,
                                     keep_tile_order=iteration > 0 or j > 0, want_loss=iteration == num_iter - 1 and j == len(ids) - 1,
                                     **({"projected": True} if (was_projected and j == 0) else {}))
                        if k in self._ba_ids:      # this view's pose gradient (its slot is overwritten by the pose's next view)
                            self._ba_grad[self._ba_ids.index(k)] += self._ba_state[k][6]
                        if len(ids) > 1:
                            if j == 0:
                                eng.acc[:14 * P].copy_(eng.flat[:14 * P])
                            else:
                                eng.acc[:14 * P].add_(eng.flat[:14 * P])
                    if len(ids) > 1:
                        eng.flat[:14 * P].copy_(eng.acc[:14 * P])
                    # how the summed gradients become stepped parameters everywhere: one flat all-reduce + the identical step on every replica, or
                    # (WindowParallel.optimizer; default from 500 k Gaussians on) reduce-scatter -> Adam on this rank's 1 / world of the elements ->
                    # all-gather of the parameters; the statistics tail and the radii always travel by all-reduce (every replica prunes)
                    shard = (not prune_now) and self.window.shard_optimizer(P)
                    if shard and self._opt_mask is not None:      # bundle adjustment: frozen Gaussians keep a zero gradient (masked before the sum: the mask is the same everywhere)
                        keep = self._opt_mask.to(eng.flat.dtype)
                        for t in eng.grads.values():
                            t.mul_(keep.view(-1, *([1] * (t.dim() - 1))))
                    if shard:
                        self._shard_buffers(eng, P)
                        self.window.reduce_scatter_flat(eng.flat, 14 * P, eng.shard_g, tail=2 * P if densify else 0, rmax=eng.stat_delta[0] if densify else None)
                    elif densify:
                        self.window.reduce_flat(eng.flat[:16 * P], eng.stat_delta[0])
                    else:
                        self.window.reduce_flat(eng.flat[:14 * P])
                    if densify:
                        torch.maximum(g.max_radii2D, eng.stat_delta[0], out=g.max_radii2D)
                        torch._foreach_add_([g.xyz_gradient_accum, g.denom], [eng.stat_delta[1], eng.stat_delta[2]])      # (one launch)
                    if self._ba_ids:
                        self._ba_window_step(eng, m, all_ids, ids)
                    if shard:
                        self._sharded_step(eng, g, P)
                    elif not prune_now:
                        fuse = (self.fuse_adam_project and iteration + 1 < num_iter and hasattr(eng, "adam_project") and eng.can_adam_project(g))
                        if fuse:
                            # the step and the NEXT view's projection + binning in one launch: the next step's keyframe picks are drawn now
                            # (same draws, same order as at the head of the next iteration), its first local view's pose goes with the step
                            pending = self.window.take_all(pop)
                            eng.adam_project(view_of(pending[1][0])[0], g, eng.grads, self._inline_adam(1))      # (opt_mask rides in the Adam struct)
                            projected_next = True
                        else:
                            if self._opt_mask is not None:       # bundle adjustment: Gaussians outside the covisible set keep a zero gradient (slam/mapper.py:931-938)
                                keep = self._opt_mask.to(eng.flat.dtype)
                                for t in eng.grads.values():
                                    t.mul_(keep.view(-1, *([1] * (t.dim() - 1))))
                            self._adam_step(eng)
                else:
                    # a pruning iteration: gradients + statistics only (the reference prunes BEFORE optimizer.step(): the
                    # parameters are replaced, so that step is a no-op)
                    stats = (g.max_radii2D, g.xyz_gradient_accum, g.denom) if densify else None
                    eng.map_loop([view_of(ids[0])], g, lcfg, stats, None, grads=eng.grads, keep_tile_order=iteration > 0,
                                 want_loss=iteration == num_iter - 1)
                if prune_now:
                    if multi:
                        self._sync_moments(eng)      # (sharded optimiser steps: the compaction moves rows across shard boundaries)
                    # on the device: predicate kernel, compaction plan, a 4-byte read-back of the new size, and -- only if
                    # something is pruned -- one scatter launch over parameters, moments and statistics (gaussian_model.py).
                    # The read-back blocks until the GPU has drained: the views of the run that follows (same keyframe picks, in the
                    # same order, whatever the pruning decides) are popped and tabulated first, while the GPU is still working.
                    if not multi and iteration + 1 < num_iter and not prune_at(iteration + 1):
                        nxt = _Views(view_of(pop()) for _ in range(run_length(iteration + 1)))
                        nxt.table = FusedEngine.view_table(nxt)
                        prepared = (iteration + 1, nxt)
                    # The capacity header (sticky: every forward since the last read, the tracker's included) rides on the pruning
                    # step's read-back: its copy to pinned memory is enqueued before, parsed after -- no round trip of its own
                    token = eng.check_capacity_begin() if (piggyback and hasattr(eng, "check_capacity_begin")) else None
                    pruned = g.prune(m["min_opacity"], self.camera_extent, None if splatam else m["size_threshold"], lazy_mask=True)
                    if token is not None:
                        if not eng.check_capacity_end(token):
                            self._late_overflow = True
                        self._piggybacked = True
                    elif piggyback and hasattr(eng, "check_capacity"):
                        if not eng.check_capacity():
                            self._late_overflow = True
                        self._piggybacked = True
                    if self._opt_mask is not None and self._opt_mask.shape[0] != g._xyz.shape[0]:
                        self._opt_mask = self._opt_mask[~pruned()].contiguous()
                iteration += 1
            if multi:
                self._sync_moments(eng)      # whoever touches the map next (seeding, snapshot, checkpoint) finds whole replicas

    def _ba_window_step(self, eng, m, all_ids, my_ids):
        """Bundle adjustment with a sharded window: this rank's views wrote their pose gradients out (Mm3dgsMapView.dpose_out_or_null)
        and the loop summed them per pose (a pose rendered twice in a step gets both gradients, like autograd accumulates them);
        here they are all-reduced over the ranks, then every replica steps -- with mm3dgs_adam: Adam(eps=1e-15), cam_q_lr / cam_t_lr,
        slam/mapper.py:742-752 -- exactly the poses SOME rank rendered in this step (known on every rank from the shared keyframe
        stack: no flags travel), so a pose nobody rendered keeps its moments and step counter, as torch skips parameters without a
        gradient (the torch-graph window does the same through WindowParallel.reduce_pose_grads)."""
        st_all, order = self._ba_state, self._ba_ids
        gbuf = self._ba_grad           # (this rank's views added their gradients as they were rendered)
        self.window.reduce_small(gbuf)
        for k in sorted(set(i for i in all_ids if i in st_all)):
            buf, mom, var, step, _ad, _pose, _slot = st_all[k]
            step[0] += 1
            gk = gbuf[order.index(k)]
            table = (_lib.Mm3dgsAdamGroup * 8)()
            for j, (lo, hi, lr) in enumerate(((0, 4, float(m["cam_q_lr"])), (4, 7, float(m["cam_t_lr"])))):
                e = table[j]
                e.param, e.grad = buf[lo:hi].data_ptr(), gk[lo:hi].data_ptr()
                e.exp_avg, e.exp_avg_sq = mom[lo:hi].data_ptr(), var[lo:hi].data_ptr()
                e.n, e.lr = hi - lo, lr
            _lib.check(eng.lib.mm3dgs_adam(table, 2, step[0], 0.9, 0.999, 1e-15, _stream()))

    def _inline_adam(self, n=1):
        """Mm3dgsMapAdam over the optimiser's own state tensors (created like torch.optim.Adam would on its first step);
        `step` = step number of the first of the n iterations it will serve (the optimiser's counters advance by n)."""
        opt = self.gaussians.optimizer
        ma = _lib.Mm3dgsMapAdam()
        step_val = None
        for i, name in enumerate(("xyz", "f_dc", "opacity", "scaling", "rotation")):
            group = next(gr for gr in opt.param_groups if gr["name"] == name)
            p = group["params"][0]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            first = int(st["step"].item()) + 1
            st["step"] += n
            step_val = first if step_val is None else step_val
            ma.param[i], ma.exp_avg[i], ma.exp_avg_sq[i] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            ma.lr[i] = float(group["lr"])
        if int(self.gaussians.active_sh_degree) > 0:      # (ABI 209) the sixth group: f_rest, stepped in the kernel at an active SH degree > 0
            group = next(gr for gr in opt.param_groups if gr["name"] == "f_rest")
            p = group["params"][0]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += n
            ma.rest_param, ma.rest_exp_avg, ma.rest_exp_avg_sq = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            ma.rest_lr = float(group["lr"])
        b1, b2 = opt.param_groups[0]["betas"]
        ma.beta1, ma.beta2, ma.eps, ma.step = float(b1), float(b2), float(opt.param_groups[0]["eps"]), step_val
        if getattr(self, "_opt_mask", None) is not None:
            ma.opt_mask = self._opt_mask.data_ptr()
        return ma

    _FLAT_GROUPS = (("xyz", 0, 3), ("f_dc", 3, 6), ("opacity", 6, 7), ("scaling", 7, 10), ("rotation", 10, 14))      # the engine's flat layout, in units of P

    def _flat_groups(self, P):
        """[(optimiser group, its parameter, its Adam state, a, b)]: the five stepped groups and their element ranges [a, b) in the
        engine's flat gradient layout."""
        opt = self.gaussians.optimizer
        out = []
        for name, a, b in self._FLAT_GROUPS:
            group = next(gr for gr in opt.param_groups if gr["name"] == name)
            p = group["params"][0]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            out.append((group, p, st, a * P, b * P))
        return out

    def _sharded_step(self, eng, g, P):
        """One optimiser step of the sharded window with the optimiser itself sharded over the ranks by ELEMENT (the
        reference's step is slam/mapper.py:931-948 on one view's gradients): the caller's reduce-scatter of the flat gradient [14 P] left
        this rank's S = ceil(14 P / world) summed elements in eng.shard_g; mm3dgs_adam on exactly those elements of the parameters and of both moments (the slice cuts
        through the five groups: one table entry per group it touches); all-gather of the stepped PARAMETERS.  The moments of elements
        another rank owns go stale here and are gathered when somebody needs them (_sync_moments: before map surgery and at the end of
        the loop).  Same arithmetic per element as the all-reduce path's mm3dgs_adam: bit-identical parameters (tested over gloo)."""
        w = self.window
        n = 14 * P
        S, lo, hi = w.shard_bounds(n)
        groups = self._flat_groups(P)
        table = (_lib.Mm3dgsAdamGroup * 8)()
        k, cuts = 0, []
        step_val = None
        for group, p, st, a, b in groups:
            st["step"] += 1                                     # (every replica counts every step of every group, as the replicated path does)
            step_val = int(st["step"].item()) if step_val is None else step_val
            i0, i1 = max(lo, a), min(hi, b)
            if i0 >= i1:
                continue
            e = table[k]; k += 1
            e.param = p.data_ptr() + 4 * (i0 - a)
            e.grad = eng.shard_g.data_ptr() + 4 * (i0 - lo)
            e.exp_avg, e.exp_avg_sq = st["exp_avg"].data_ptr() + 4 * (i0 - a), st["exp_avg_sq"].data_ptr() + 4 * (i0 - a)
            e.n, e.lr = i1 - i0, float(group["lr"])
            cuts.append((p, a, i0, i1))
        opt = g.optimizer
        b1, b2 = opt.param_groups[0]["betas"]
        if k:
            _lib.check(eng.lib.mm3dgs_adam(table, k, step_val, float(b1), float(b2), float(opt.param_groups[0]["eps"]), _stream()))
        for p, a, i0, i1 in cuts:
            eng.shard_p[i0 - lo:i1 - lo].copy_(p.detach().view(-1)[i0 - a:i1 - a])
        w.all_gather_flat(eng.pflat, eng.shard_p, n)
        for group, p, st, a, b in groups:
            p.detach().view(-1).copy_(eng.pflat[a:b])
        self._moments_stale = True
        w.sharded_steps += 1

    def _shard_buffers(self, eng, P):
        w = self.window
        S = w.shard_bounds(14 * P)[0]
        if getattr(eng, "_shard_S", None) != (S, w.world):
            eng._shard_S = (S, w.world)
            eng.shard_g = torch.zeros(S, device=eng.dev)
            eng.shard_p = torch.zeros(S, device=eng.dev)
            eng.pflat = torch.zeros(w.world * S, device=eng.dev)

    def _sync_moments(self, eng):
        """After sharded steps every rank holds fresh Adam moments for ITS elements only: gather both moment arrays so that the replicas
        are whole again (before a pruning step compacts rows across shard boundaries, before a snapshot / checkpoint, at the end of the
        loop).  Two all-gathers of 14 P floats, a few times per frame."""
        if not getattr(self, "_moments_stale", False):
            return
        self._moments_stale = False
        g, w = self.gaussians, self.window
        P = int(g._xyz.shape[0])
        n = 14 * P
        S, lo, hi = w.shard_bounds(n)
        self._shard_buffers(eng, P)      # (sized at the last sharded step: the map may have changed size since -- ADVICE round 5)
        assert eng._shard_S == (S, w.world) and eng.pflat.numel() >= w.world * S
        groups = self._flat_groups(P)
        for key in ("exp_avg", "exp_avg_sq"):
            for group, p, st, a, b in groups:
                i0, i1 = max(lo, a), min(hi, b)
                if i0 < i1:
                    eng.shard_p[i0 - lo:i1 - lo].copy_(st[key].view(-1)[i0 - a:i1 - a])
            w.all_gather_flat(eng.pflat, eng.shard_p, n)
            for group, p, st, a, b in groups:
                st[key].view(-1).copy_(eng.pflat[a:b])

    def _adam_step(self, eng):
        g = self.gaussians
        opt = g.optimizer
        # the group table only changes when a tensor is replaced (prune / densify): cache it on the pointers
        key = (g.generation, g._xyz.data_ptr(), eng.grads["xyz"].data_ptr(), int(g._xyz.shape[0]))
        cache = getattr(self, "_adam_cache", None)
        if cache is not None and cache[0] == key and all(float(gr["lr"]) == lr for gr, lr in zip(cache[3], cache[4])):
            _, table, n, groups, _, states = cache
            for st in states:
                st["step"] += 1
            step_val = int(states[0]["step"].item())
            b1, b2 = opt.param_groups[0]["betas"]
            _lib.check(eng.lib.mm3dgs_adam(table, n, step_val, float(b1), float(b2), float(opt.param_groups[0]["eps"]), _stream()))
            return
        table = (_lib.Mm3dgsAdamGroup * 8)()
        n = 0
        step_val = None
        used_groups, used_states = [], []
        for group in opt.param_groups:
            name = group["name"]
            if name not in eng.grads:
                continue                     # f_rest (empty at SH degree 0) and rgb never receive a gradient
            p = group["params"][0]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            step_val = int(st["step"].item()) if step_val is None else step_val
            e = table[n]
            e.param, e.grad = p.data_ptr(), eng.grads[name].data_ptr()
            e.exp_avg, e.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            e.n, e.lr = p.numel(), float(group["lr"])
            used_groups.append(group); used_states.append(st)
            n += 1
        self._adam_cache = (key, table, n, used_groups, [float(gr["lr"]) for gr in used_groups], used_states)
        b1, b2 = opt.param_groups[0]["betas"]
        _lib.check(eng.lib.mm3dgs_adam(table, n, step_val, float(b1), float(b2), float(opt.param_groups[0]["eps"]), _stream()))
