/*
 * mm3dgs.h -- C ABI of libmm3dgs_hip.so: the MI355X (gfx950) differentiable 3D-Gaussian tile rasterizer that
 * sits behind MM3DGS-SLAM's render boundary.
 *
 * What this boundary replaces in the reference (/root/reference):
 *   - the un-vendored CUDA extension `diff_gaussian_rasterization` (.gitmodules:1-3; imported only at
 *     slam/renderer.py:15-18).  Its Python surface -- GaussianRasterizationSettings (slam/renderer.py:125-138)
 *     and GaussianRasterizer.__call__ (slam/renderer.py:140,196-214) plus the autograd backward driven by
 *     slam/tracker.py:157 and slam/mapper.py:875 -- is rebuilt in mm3dgs_slam_amd/rasterizer.py on top of the
 *     entry points below through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with `host_`; all buffers are caller-owned
 *     (torch allocations); the library never allocates device memory.
 *   - float32 everywhere, row-major; matrices use the reference's row-vector convention
 *     (p_view = [p,1] * viewmatrix, slam/renderer.py:117-124).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no entry point synchronises the
 *     host.  A caller that wants an exactly-sized binning buffer runs stage 1, waits for `host_num_rendered`
 *     itself, then runs stage 2 (this is what the Python shim's "exact" policy does).
 *   - return value: 0 on success, negative on error; mm3dgs_last_error() returns a thread-local message.
 */
#ifndef MM3DGS_H
#define MM3DGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM3DGS_TILE 16
#define MM3DGS_MAX_CHANNELS 6
#define MM3DGS_SPLAT_FLOATS 12 /* packed per-Gaussian screen-space record: xy, conic(3), opacity, 6 colours */

/* Per-call camera / configuration bundle == GaussianRasterizationSettings (slam/renderer.py:125-138). */
typedef struct Mm3dgsCamera {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;          /* W / (2 fx), slam/renderer.py:61 */
  float tanfovy;          /* H / (2 fy), slam/renderer.py:62 */
  float scale_modifier;
  int32_t sh_degree;      /* active degree, 0..3 */
  int32_t prefiltered;
  int32_t debug;
  const float* bg;         /* [3]   background, added as T_final * bg (slam/renderer.py:80-83,130) */
  const float* viewmatrix; /* [4,4] row-vector convention (w2c^T), or identity in transform_means_python mode */
  const float* projmatrix; /* [4,4] viewmatrix @ P^T (slam/renderer.py:121-123) */
  const float* campos;     /* [3] */
} Mm3dgsCamera;

/* Header at offset 0 of the image-state buffer; counters are written by the device. */
typedef struct Mm3dgsHeader {
  uint32_t num_rendered; /* N = sum of tiles touched (the reference lineage's `num_rendered`) */
  uint32_t overflow;     /* 1 if N exceeded the binning capacity of a call (output then incomplete).  STICKY in the fused
                            SLAM path (persistent state, MM3DGS_FWD_STATE_CLEAN): set by any forward since the host last
                            cleared it, so one read after a whole optimisation loop sees an overflow of ANY iteration.
                            While it is set the SLAM backward entry points take NO optimiser step (map Adam, pose Adam), write
                            zero gradients and leave the densification statistics alone: an overflowing forward dropped pairs
                            without writing their gradient records, so its iteration -- and every later one of the loop, until
                            the host clears the word -- is void rather than wrong */
  uint32_t max_tile_len; /* longest per-tile list (fused SLAM path: maximum since the host last cleared it) */
  uint32_t max_num_rendered; /* fused SLAM path: maximum N since the host last cleared it (capacity model) */
  /* diagnostics, only counted when the environment variable MM3DGS_STATS=1 (adds atomics; not for timing runs) */
  uint32_t fwd_wave_iters; /* (wave, splat) evaluations executed by the forward compositor  */
  uint32_t bwd_wave_iters; /* (wave, splat) evaluations that reached the gradient reduction */
  uint32_t bwd_wave_visits;/* (wave, splat) evaluations executed by the backward compositor */
  uint32_t bin_cap;        /* 0: the tile bins are packed (bin of tile t = [ranges[t], ranges[t+1])); otherwise every tile owns a
                              fixed span of bin_cap pairs starting at t * bin_cap (MM3DGS_FWD_DIRECT_BINS) and ranges[t] holds
                              its length.  Written by every forward. */
  uint32_t max_group_records; /* direct bins: most gradient records of one projection workgroup (256 Gaussians) since the host
                                 last cleared it (sticky); every workgroup owns 16 * N_capacity / ceil(P / 256) records of the
                                 backward scratch, a workgroup that needs more sets `overflow` */
  uint32_t tile_order_tiles;  /* 0: workgroup -> tile by arithmetic; H << 16 | W: image_state holds a load-balanced workgroup -> tile table
                                 for an H x W image, written by the SLAM loop entry points from the list lengths of the last render
                                 (any permutation of the tiles renders the same image; only the speed depends on it) */
  uint32_t overflow_seen;     /* ABI 205: the value of `overflow` as the mapping-mode backward compositor of an iteration found it, i.e. after
                                 every launch of THAT iteration's forward.  The fused backward projection + next projection launch reads
                                 this copy: its second half bins the NEXT view and may raise `overflow` while workgroups of its first half
                                 are still starting -- with the copy, all of them take the same step / no-step decision */
  uint32_t mean_wave_steps;   /* ABI 207: mean list walk of a compositing wave (8x8 sub-tile) in the render the workgroup -> tile table was built from
                                 (written with the table, 0: unknown): the compositors raise the issue priority of waves whose walk is well above it */
} Mm3dgsHeader;

struct Mm3dgsLossConfig;   /* defined with mm3dgs_loss below; the SLAM loop entry points take a pointer to it */

/* ---- buffer sizing (pure host arithmetic) ------------------------------------------------------------- */
size_t mm3dgs_geom_bytes(int P);                       /* per-Gaussian screen-space state               */
size_t mm3dgs_image_bytes(int H, int W);               /* header + per-tile counters/ranges + per-pixel */
size_t mm3dgs_binning_bytes(size_t N_capacity);        /* (depth,id) keys + 4x4-block lists + block masks */
size_t mm3dgs_backward_scratch_bytes(int P, size_t N_capacity); /* per-(sub-tile,splat) gradient records + camera */

/* ---- forward ------------------------------------------------------------------------------------------
 * Channels: C = n_sh_channels + n_extra.  If `shs` != NULL the first 3 channels are SH colour evaluated at
 * cam->sh_degree with M coefficients per Gaussian ([P,M,3], slam/renderer.py:191) and `colors_precomp`
 * (may be NULL) supplies `C-3` extra channels [P,C-3]; otherwise `colors_precomp` supplies all C channels
 * [P,C] (slam/renderer.py:207-214 passes [z,1,z^2]).  C in {1..6}.  Extra channels beyond 3 use bg = 0.
 * Exactly one of (scales,rotations) / cov3D_precomp ([P,6]) must be non-NULL.
 *
 * Stage 1: project, cull, EWA-splat to 2D conics, colour, count per-tile overlaps, scan tile counts.
 *          Writes radii[P] (int32, 0 == culled) and the header (num_rendered).  If host_num_rendered is
 *          non-NULL (must be pinned host memory) N is also copied there asynchronously. */
int mm3dgs_forward_geom(const Mm3dgsCamera* cam, int P, int M, int C, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_state,
                        void* image_state, uint32_t* host_num_rendered, void* stream);

/* Stage 2: scatter (depth,id) keys into per-tile bins, sort every tile's bin in LDS, composite front to back.
 *          Writes out_color[C,H,W].  N_capacity is the number of (tile,Gaussian) pairs `binning_state` was
 *          sized for; on overflow header.overflow is set and the image is incomplete (never out of bounds). */
int mm3dgs_forward_raster(const Mm3dgsCamera* cam, int P, int C, const void* geom_state, void* image_state,
                          void* binning_state, size_t N_capacity, float* out_color, void* stream);

/* Both stages back to back, no host synchronisation (capacity-bounded binning). */
int mm3dgs_forward(const Mm3dgsCamera* cam, int P, int M, int C, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales,
                   const float* rotations, const float* cov3D_precomp, float* out_color, int32_t* radii,
                   void* geom_state, void* image_state, void* binning_state, size_t N_capacity, void* stream);

/* ---- backward -----------------------------------------------------------------------------------------
 * dL_dout [C,H,W] -> gradients of every tensor input of the forward.  Output pointers may be NULL when that
 * gradient is not wanted.  dL_dmeans2D is [P,3] with (x,y) = dL/d(ndc-scaled screen position), z = 0, which
 * is what slam/gaussian_model.py:594-598 consumes.  Camera gradients dL_dview[16], dL_dproj[16],
 * dL_dcampos[3] (row-vector layout, same as the inputs) are produced when non-NULL ("-w-pose" behaviour,
 * slam/renderer.py:115-124).  flags: see MM3DGS_BWD_*. */
#define MM3DGS_BWD_SKIP_GAUSSIAN_GRADS 1 /* tracker mode: only dL_dmeans3D / dL_dcolors_precomp / camera */

int mm3dgs_backward(const Mm3dgsCamera* cam, int P, int M, int C, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* opacities, const float* scales,
                    const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                    const void* geom_state, const void* image_state, const void* binning_state,
                    size_t N_capacity, const float* dL_dout, void* backward_scratch, float* dL_dmeans3D,
                    float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors_precomp, float* dL_dopacities,
                    float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dview, float* dL_dproj,
                    float* dL_dcampos, int flags, void* stream);

/* visible[P] (uint8) = z_view > 0.2 -- the lineage's markVisible. */
int mm3dgs_mark_visible(const Mm3dgsCamera* cam, int P, const float* means3D, uint8_t* visible, void* stream);

/* =====================================================================================================
 * Fused SLAM iteration (section 8f rows 1-2).
 * One projection kernel folds in the pose transform of `transform_means_python` mode (slam/renderer.py:142-153),
 * the [z,1,z^2] depth bundle (:26-43), the GaussianModel activations (slam/gaussian_model.py:108-137) and, in the
 * backward, their chain rules, the pose gradient (P -> 12 float reduction -> (dq,dt)), the densification statistics
 * (slam/mapper.py:887-899) and the pose Adam step (slam/tracker.py:233-246).  SH degree 0 only (both shipped configs).
 * Output image has 6 channels: RGB, alpha-weighted z, silhouette, alpha-weighted z^2.
 * ===================================================================================================== */
typedef struct Mm3dgsSlamInputs {
  const float* pose;      /* [7] (qw,qx,qy,qz,tx,ty,tz) world->camera, raw                              */
  const float* xyz;       /* [P,3] GaussianModel._xyz                                                    */
  const float* f_dc;      /* [P,3] GaussianModel._features_dc ([P,1,3])                                  */
  const float* opacity;   /* [P]   logits (GaussianModel._opacity)                                       */
  const float* scaling;   /* [P,3] log-scales                                                            */
  const float* rotation;  /* [P,4] raw quaternions (w,x,y,z)                                             */
  int32_t isotropic;      /* pipeline.force_isotropic (slam/renderer.py:167-168)                         */
  int32_t world_means;    /* 0: pipeline.transform_means_python (the shipped configs): the means are moved to the camera frame, the
                             rasterizer's view matrix is the identity and the Gaussians' covariances keep their WORLD orientation
                             (slam/renderer.py:142-153,171-173); 1: transform_means_python: false (:117-124): world-frame means under
                             the full view matrix -- the covariance is rotated into the view, the pose gradient also flows through that
                             rotation, and the depth bundle takes the reference's literal z' = (w2c^T [x; 1])_z of :207-214 (the
                             transposed matrix: third COLUMN of R, no translation)                          */
  /* ABI 209 (appended): an ACTIVE spherical-harmonics degree above 0 -- a map resumed from a checkpoint starts at its maximal degree
   * (slam/gaussian_model.py:363), and slam/renderer.py:179-193 then hands the rasterizer shs = cat(f_dc, f_rest) with sh_degree = the active
   * degree and campos = 0 in the shipped mode (means pre-transformed: the viewing direction is the camera-space mean, normalised).  NULL / 0:
   * degree 0 (colour = SH_C0 f_dc + 0.5, no direction).  world_means = 1 with sh_degree > 0 is not supported (error -2).        */
  const float* f_rest;    /* [P, n_rest, 3] GaussianModel._features_rest (n_rest = (max_sh_degree + 1)^2 - 1) or NULL    */
  int32_t sh_degree;      /* active degree 0..3; its (sh_degree + 1)^2 - 1 first rows of f_rest are used            */
  int32_t n_rest;         /* rows of f_rest per Gaussian                                                            */
} Mm3dgsSlamInputs;

typedef struct Mm3dgsSlamGrads {
  float* d_xyz; float* d_f_dc; float* d_opacity; float* d_scaling; float* d_rotation; /* all or none (tracking) */
  float* max_radii2D; float* grad_accum; float* denom; /* [P] updated in place when max_radii2D != NULL          */
  float* d_f_rest;        /* ABI 209 (appended): [P, n_rest, 3] with the other d_* when Mm3dgsSlamInputs.sh_degree > 0 (rows beyond the active degree: 0) */
} Mm3dgsSlamGrads;

/* Optional: the map's Adam step applied inside the backward projection kernel (no gradient round trip through HBM, one
 * launch less).  Same formula as mm3dgs_adam / torch.optim.Adam; groups in the order xyz, f_dc, opacity, scaling, rotation.
 * When passed to mm3dgs_slam_backward the d_* outputs of Mm3dgsSlamGrads may be NULL. */
typedef struct Mm3dgsMapAdam {
  float* param[5]; float* exp_avg[5]; float* exp_avg_sq[5];
  /* hyper-parameters are doubles, exactly the Python floats torch.optim.Adam holds: the library forms 1 - beta, the bias
   * corrections and lr / (1 - beta1^step) in double and rounds ONCE to float, like torch does (1 - 0.999f in float arithmetic
   * is off by 1.3e-5 relative) */
  double lr[5]; double beta1, beta2, eps; int32_t step;
  /* optional: uint8 [P]; the gradient of a Gaussian with opt_mask == 0 is set to zero before the step (bundle adjustment optimises
   * the Gaussians seen by >= 2 window keyframes plus the new ones, slam/mapper.py:931-936; a zero gradient still decays the moments,
   * as `p.grad[~mask] = 0` followed by optimizer.step() does).  NULL: every Gaussian. */
  const uint8_t* opt_mask;
  /* ABI 209 (appended): the sixth group, f_rest [P, n_rest, 3] (lr = feature_lr / 20, slam/gaussian_model.py:143-195), stepped in the kernel when
   * Mm3dgsSlamInputs.sh_degree > 0 (rows beyond the active degree take a zero gradient: moments decay, as torch's Adam does it); NULL otherwise */
  float* rest_param; float* rest_exp_avg; float* rest_exp_avg_sq; double rest_lr;
} Mm3dgsMapAdam;

typedef struct Mm3dgsPoseAdam { /* torch.optim.Adam on (q; lr_q) and (t; lr_t); pose == NULL: no step */
  float* pose; float* m; float* v; int32_t* step; double lr_q, lr_t, beta1, beta2, eps;   /* doubles: see Mm3dgsMapAdam */
  /* optional IMU relative-pose residual added to the tracking loss (utils/loss_utils.py:20-40 rel_pose_loss, used at
   * slam/tracker.py:146-155): w_t * |t - t0|^2 + w_q * 2 acos(|normalize(q (x) conj(q0))_w|), (q0, t0) = prior_pose[7]
   * (device; the pose the optimisation started from).  NULL or both weights 0: no residual.  At q == q0 the angle term's
   * autograd gradient is NaN in the reference (acos'(1) = -inf times 0); here it is taken as 0 there. */
  const float* prior_pose; float prior_w_t, prior_w_q;
  /* ABI 206, optional: the best pose candidate of the loop (slam/tracker.py:88-91,161-181 -- the reference computes it and, by a
   * rebinding bug, discards it; this repository's `keep_best_candidate` option keeps it).  best[8] = { loss, pose[7] }, initialised by the
   * caller to { 1e20, starting pose }: after every step, if the iteration's total loss (image terms + prior; evaluated at the pose BEFORE
   * the step, as the reference compares it) is below best[0], best <- { loss, the pose AFTER the step }.  NULL: not tracked. */
  float* best;
} Mm3dgsPoseAdam;

/* The SLAM entry points (mm3dgs_slam_*) render the reference's bundle -- RGB and, as channels 3..5, the depth pass [z, 1, z^2] of
 * slam/renderer.py:207-214 -- in one pass.  The reference composites BOTH passes over cam->bg (renderer.py:80-83,196-214), so here
 * channels 3..5 receive T_final * bg[ch - 3] as well (with `white_background` the silhouette channel is 1 everywhere), and the backward
 * carries the corresponding -T_final bg . dL term.  (mm3dgs_forward with C = 6 keeps "extra channels over black", see above.) */
/* flags for mm3dgs_slam_forward */
#define MM3DGS_FWD_STATE_CLEAN 1 /* image_state's header+tile counters are already zero (the library leaves them zero
                                    after every forward), so the per-call memset is skipped: for persistent state buffers */
#define MM3DGS_FWD_DIRECT_BINS 4 /* N_capacity was sized as T x (per-tile capacity): projection and binning run as ONE launch that
                                    drops every (Gaussian, tile) pair straight into the tile's fixed span (no tile counting pass, no
                                    scan).  A tile with more pairs than N_capacity / T sets `overflow`.  Needs STATE_CLEAN and
                                    SHORT_LISTS; ignored (packed bins) otherwise, or when the map is too large for the key layout
                                    (id and slot share 32 bits: spans of up to 8191 pairs to 512 k Gaussians, 4095 at 1 M, none
                                    beyond 4 M).  The gradient records are
                                    then laid out per projection workgroup (see Mm3dgsHeader.max_group_records). */
#define MM3DGS_FWD_SHORT_LISTS 2 /* hint: no tile list exceeds 2048 splats -> one sort launch (longer lists stay correct
                                    through the global-memory path, only slower)                                        */
#define MM3DGS_FWD_PROJECTED 16 /* mm3dgs_slam_forward / the FIRST view of mm3dgs_slam_map: projection + binning of this view were
                                    already launched by mm3dgs_slam_adam_project (direct bins only; ignored otherwise)            */
#define MM3DGS_FWD_KEEP_TILE_ORDER 8 /* mm3dgs_slam_track / mm3dgs_slam_map: do not rebuild image_state's load-balanced workgroup -> tile
                                    table at the head of this call (Mm3dgsHeader.tile_order_tiles); the table of an earlier call stays
                                    in force while it matches the image size.  For callers that enqueue ONE iteration per call (the
                                    multi-GPU window: gradients out, all-reduce, step): any valid table renders the same image, a
                                    fresher one only balances better, and rebuilding it costs a 9 us launch                 */
int mm3dgs_slam_forward(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, float* out_color /*[6,H,W]*/,
                        int32_t* radii, void* geom_state, void* image_state, void* binning_state, size_t N_capacity,
                        int flags, void* stream);
int mm3dgs_slam_backward(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, const int32_t* radii,
                         const void* geom_state, const void* image_state, const void* binning_state, size_t N_capacity,
                         const float* dL_dout /*[6,H,W]*/, void* backward_scratch, const Mm3dgsSlamGrads* grads,
                         float* dL_dpose /*[7] or NULL*/, const Mm3dgsPoseAdam* pose_adam, const Mm3dgsMapAdam* map_adam,
                         int flags /* the flags of the mm3dgs_slam_forward call whose state this is */, void* stream);

/* Visibility of the map from one pose with the projection stage alone (no binning, no compositing): radii[P] as the forward pass
 * would report them, and seen_count[i] += (radii[i] > 0) when seen_count != NULL -- what get_covisible_gaussians needs
 * (slam/mapper.py:690-716: Gaussians visible in >= 2 window keyframes).  geom_state: scratch of mm3dgs_geom_bytes(P). */
int mm3dgs_slam_visibility(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, int32_t* radii, uint32_t* seen_count_or_null,
                           void* geom_state, void* stream);

/* n_iter tracking iterations enqueued back to back from C (slam/tracker.py:94-177 with the "vigs" loss): each is
 * forward -> loss -> backward with the pose Adam step on the device; the pose buffer is updated in place. */
int mm3dgs_slam_track(int n_iter, const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, float* out_color,
                      int32_t* radii, void* geom_state, void* image_state, void* binning_state, size_t N_capacity,
                      int fwd_flags, const struct Mm3dgsLossConfig* loss_cfg, const float* gt_color,
                      const float* ref_depth_or_null, void* loss_work, float* dL_dout, float* loss4,
                      void* backward_scratch, const Mm3dgsPoseAdam* pose_adam, void* stream);

/* n_iter mapping iterations enqueued back to back from C (the loop of slam/mapper.py:803-950 between two pruning steps):
 * iteration i renders view i of the window (its pose, colour target, optional reference depth), takes the mapping loss and
 * the backward pass with the map's Adam step inside (map_adam->step = step number of iteration 0, +1 per iteration);
 * `grads_stats` (may be NULL): max_radii2D / grad_accum / denom = the densification statistics of slam/mapper.py:887-899;
 * its d_* pointers, when set, receive the parameter gradients of the LAST iteration (map_adam may then be NULL: a window
 * rank that all-reduces gradients before a common Adam step runs n_iter = 1 this way).  `in->pose` is ignored.  No host
 * synchronisation; `views` is read on the host during the call. */
typedef struct Mm3dgsMapView {
  const float* pose; const float* gt_color; const float* ref_depth_or_null;
  /* bundle adjustment (mapping.do_BA, slam/mapper.py:718-795,931-942): when set, the pose gradient of this view is reduced and
   * the Adam step of ITS pose is taken on the device (pose_adam->pose must be `pose`; per-keyframe moments and step counter). */
  const Mm3dgsPoseAdam* pose_adam_or_null;
  /* bundle adjustment with a sharded mapping window (pose gradients are summed over the ranks before a common step):
   * when set (and pose_adam_or_null is NULL), the view's pose gradient [7] = dL/d(qw, qx, qy, qz, tx, ty, tz) is written here and
   * no pose step is taken. */
  float* dpose_out_or_null;
} Mm3dgsMapView;
/* (loss4 may be NULL: the loss scalars of the call's last iteration are then not finished -- one launch less) */
int mm3dgs_slam_map(int n_iter, const Mm3dgsMapView* views, const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in,
                    float* out_color, int32_t* radii, void* geom_state, void* image_state, void* binning_state,
                    size_t N_capacity, int fwd_flags, const struct Mm3dgsLossConfig* loss_cfg, void* loss_work, float* dL_dout,
                    float* loss4, void* backward_scratch, const Mm3dgsSlamGrads* grads_stats, const Mm3dgsMapAdam* map_adam,
                    void* stream);

/* Image losses with the gradient image as output (slam/tracker.py:104-155, slam/mapper.py:836-873,
 * utils/loss_utils.py): w_l1 * mean|rgb-gt| (optionally over silhouette > sil_thr) + w_ssim * (1 - SSIM 11x11)
 * + w_pearson * (1 - rho(depth, ref)).  loss[4] = {total, l1, 1-ssim, 1-rho}.  work: mm3dgs_loss_work_bytes().
 * The `method: splatam` losses (slam/tracker.py:110-126, slam/mapper.py:836-855) add a depth term and masked sums:
 *   + w_depth_l1 * mean|ref - depth| over depth_l1_mask;  l1_sum = 1 turns both L1 means into sums over their masks
 * (tracking: l1_mask = depth_l1_mask = 3, l1_sum = 1, w_l1 = 0.5, w_depth_l1 = 1;  mapping: depth_l1_mask = 2, w_depth_l1 = 1,
 * w_l1 = 0.5 (1 - lambda), w_ssim = 0.5 lambda).  The depth-L1 and Pearson terms are exclusive (they share the partial-sum
 * columns); with w_depth_l1 != 0, loss[3] is the depth term.  The reference's NaN masks are not modelled: this renderer
 * produces no NaN.  Configurations that use the three splatam fields run the loss as its own launches (not folded into the
 * compositors). */
typedef struct Mm3dgsLossConfig {
  int32_t H, W;
  float w_l1, w_ssim, w_pearson;
  int32_t l1_mask;        /* bit0 silhouette > sil_thr, bit1 ref > 0 (0: all pixels)                     */
  int32_t pearson_mask;   /* bit0 silhouette > sil_thr, bit1 ref > 0                                     */
  int32_t pearson_invert; /* 1: min over targets -ref and 1/(ref+200) (utils/loss_utils.py:53-57)        */
  float sil_thr;
  float window[11];       /* normalised 1-D Gaussian window (sigma 1.5), as utils/loss_utils.py:95-112    */
  float w_depth_l1;       /* weight of the depth-L1 term (0: none)                                       */
  int32_t depth_l1_mask;  /* bit0 silhouette > sil_thr, bit1 ref > 0                                     */
  int32_t l1_sum;         /* 1: the colour and depth L1 terms are sums over their masks, not means       */
} Mm3dgsLossConfig;
size_t mm3dgs_loss_work_bytes(int H, int W);
int mm3dgs_loss(const Mm3dgsLossConfig* cfg, const float* out6, const float* gt_color, const float* ref_depth_or_null,
                void* work, float* dL_dout6, float* loss4, void* stream);

/* Fused Adam over up to 8 parameter groups in one launch (slam/gaussian_model.py:143-195; torch.optim.Adam formula).
 * step = 1-based step count used for the bias corrections. */
typedef struct Mm3dgsAdamGroup { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; uint64_t n; double lr; } Mm3dgsAdamGroup;
int mm3dgs_adam(const Mm3dgsAdamGroup* groups, int n_groups, int step, double beta1, double beta2, double eps, void* stream);

/* The multi-GPU mapping window's optimiser step AND the next view's projection + binning in one launch: the map's Adam step
 * (slam/mapper.py:931-948) from gradient ARRAYS -- grads->d_* as mm3dgs_slam_map wrote them out and the caller all-reduced them; the
 * statistics pointers are ignored -- with `adam` as for mm3dgs_slam_backward (opt_mask included), then, from the stepped parameters
 * still in registers, what the head of the next mm3dgs_slam_map / mm3dgs_slam_forward call would launch for the view at in->pose.
 * That call must carry MM3DGS_FWD_PROJECTED (and the same cam, P, buffers, capacity and flags).  Replaces mm3dgs_adam + the next
 * call's projection launch; same arithmetic as both.  Needs direct bins (MM3DGS_FWD_DIRECT_BINS in fwd_flags and honoured for this
 * P / capacity): returns -3 otherwise, nothing done. */
/* 1 if the SLAM entry points run with direct bins for this camera, map size, capacity and flags, 0 otherwise (pure host arithmetic:
 * what a caller of mm3dgs_slam_adam_project asks first). */
int mm3dgs_slam_direct_bins(const Mm3dgsCamera* cam, int P, size_t N_capacity, int fwd_flags);
int mm3dgs_slam_adam_project(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, const Mm3dgsSlamGrads* grads,
                             const Mm3dgsMapAdam* adam, int32_t* radii, void* geom_state, void* image_state, void* binning_state,
                             size_t N_capacity, int fwd_flags, void* stream);

/* =====================================================================================================
 * Map surgery on the device: pruning predicate, order-preserving compaction of the
 * per-Gaussian SoA state, seeding of new Gaussians.  Replaces the torch nonzero / index_select / cat machinery of
 * slam/gaussian_model.py:380-451 (prune_points, _prune_optimizer) and slam/mapper.py:409-493,644-668 (get_pointcloud +
 * initialisation of the new rows).  All asynchronous; the counts land in device words the caller reads when it chooses to.
 * ===================================================================================================== */
/* keep[i] = 0 where  sigmoid(opacity[i]) < min_opacity  ||  max_k exp(log_scales[i,k]) > max_scale  ||
 * (max_radii2D != NULL && max_radii2D[i] > max_screen_size)   (slam/gaussian_model.py:574-588), else 1;
 * *n_pruned_accum += number of zeros written (a sticky counter: never reset by the library). */
int mm3dgs_prune_mask(int P, const float* opacity_logits, const float* log_scales, const float* max_radii2D_or_null, float min_opacity,
                      float max_scale, float max_screen_size, uint8_t* keep, uint32_t* n_pruned_accum, void* stream);
/* Ranks of the kept elements of keep[0..n) in order: `work` (mm3dgs_compact_work_bytes(n)) receives the plan, *n_keep the total. */
size_t mm3dgs_compact_work_bytes(size_t n);
int mm3dgs_compact_plan(size_t n, const uint8_t* keep, void* work, uint32_t* n_keep, void* stream);
/* dst[rank(i)][0..width) = src[i][0..width) for every kept i, for up to 32 arrays in one launch (rows are `width` floats). */
typedef struct Mm3dgsCompactArray { const float* src; float* dst; int32_t width; } Mm3dgsCompactArray;
int mm3dgs_compact_rows(size_t n, const uint8_t* keep, const void* work, const Mm3dgsCompactArray* arrays, int n_arrays, void* stream);
/* One new Gaussian per kept pixel (raster order) of an RGB-D frame, written to rows row0 + rank of the output arrays:
 * xyz = back-projection through `pose` (world->camera 7-vector), log-scale log(z / ((fx+fy)/2)) on all three axes, opacity logit 0,
 * identity quaternion, f_dc = (rgb - 0.5) / C0, f_rest = 0 (n_rest coefficients), rgb = colour.  Plan from mm3dgs_compact_plan(H*W). */
typedef struct Mm3dgsSeedOutputs { float* xyz; float* f_dc; float* f_rest; float* opacity; float* scaling; float* rotation; float* rgb; } Mm3dgsSeedOutputs;
int mm3dgs_seed_gaussians(int H, int W, const float* color /*[3,H,W]*/, const float* depth /*[H,W]*/, const uint8_t* keep, const void* work,
                          const float* pose, float fx, float fy, float cx, float cy, uint32_t row0, const Mm3dgsSeedOutputs* out, int n_rest,
                          void* stream);

/* Keyframe test (slam/mapper.py:141-216: need_new_keyframe -> get_depth_pointcloud + is_covisible): back-project the last keyframe's
 * rendered surface (depth where silhouette > 0.99, minus points that round to the world origin at 4 decimals) to the world and
 * project it into the current view.  counts[0] = points inside the image in front of the camera, counts[1] = points tested
 * (device memory, zeroed by the call); the covisibility ratio is counts[0] / max(counts[1], 1).  Poses are world->camera
 * (qw,qx,qy,qz,tx,ty,tz) on the device. */
int mm3dgs_covisibility_ratio(int H, int W, const float* depth /*[H,W]*/, const float* silhouette /*[H,W]*/, const float* keyframe_pose,
                              const float* current_pose, float fx, float fy, float cx, float cy, uint32_t* counts /*[2]*/, void* stream);

/* Constant-velocity pose prediction of the tracker (utils/pose_utils.py:203-216 propagate_const_vel, called at slam/tracker.py:200-206):
 * out = pose of (W1 W2^-1) W1 for the world->camera poses of the last two frames, all three (qw,qx,qy,qz,tx,ty,tz) on the device;
 * computed in double precision by one lane, so the frame's first tracking launch needs no pose on the host. */
int mm3dgs_propagate_const_vel(const float* pose_m1 /*[7], frame idx-1*/, const float* pose_m2 /*[7], frame idx-2*/, float* out_pose /*[7]*/,
                               void* stream);

/* ---- optional per-kernel timing (HIP events recorded on the caller's stream around each launch) ------------
 * Used by bench.py's roofline leg.  mm3dgs_profile_read() waits for the recorded events, returns the number of
 * (timed) launches and their summed duration since the previous read, and resets the counters. */
#define MM3DGS_PROF_PREPROCESS_FWD 0
#define MM3DGS_PROF_SCAN 1
#define MM3DGS_PROF_BIN_SORT 2
#define MM3DGS_PROF_COMPOSITE_FWD 3
#define MM3DGS_PROF_COMPOSITE_BWD 4
#define MM3DGS_PROF_PREPROCESS_BWD 5
#define MM3DGS_PROF_LOSS 6
#define MM3DGS_PROF_ADAM 7
#define MM3DGS_PROF_COMPOSITE_BWD_TRACK 8 /* the tracking-mode backward compositor of the fused SLAM path (COMPOSITE_BWD: every other form) */
#define MM3DGS_PROF_TRACK_FWD_BWD 9 /* sort + forward + backward compositing of a tracking iteration in one launch */
#define MM3DGS_PROF_KERNELS 10
void mm3dgs_profile_enable(int mode); /* 0 off, 1 every kernel, 2 every 64th launch of the forward (sort +) compositor and of the backward compositors only */
int mm3dgs_profile_read(int kernel, uint64_t* launches, double* total_ms);
/* what the event pair adds to an interval it brackets (measured on empty kernels: 2 T(1 launch) - T(2 launches)); synchronises the stream */
double mm3dgs_profile_event_overhead_ms(void* stream);

/* ---- Environment (developer switches) ------------------------------------------------------------------------
 * The library reads these variables; none is needed in production, all default to "off" / the value given.  They select an alternative
 * path that the tests hold equal to the default one (bit for bit where noted) or parametrise an A/B experiment.  The first SLAM entry
 * point of a process prints one line to stderr for every one that is set.
 *   MM3DGS_NO_DIRECT_BINS=1    packed bins (count, scan, scatter) instead of direct bins in the SLAM entry points          (bit-identical)
 *   MM3DGS_NO_FUSED_SORT=1     the per-tile sort as launches of its own instead of inside the forward compositor
 *   MM3DGS_NO_FUSED_SCAN=1     a separate scan_tiles launch on the packed path (also disables direct bins)
 *   MM3DGS_NO_FUSED_TRACK=1    mm3dgs_slam_track: forward and backward compositor as two launches                            (bit-identical)
 *   MM3DGS_NO_FOLDED_LOSS=1    the loss kernels as launches of their own instead of folded into the compositors
 *   MM3DGS_NO_FORWARD_ROWS=1   mm3dgs_slam_map: the standalone mm3dgs_loss instead of the forward compositor's row sums
 *   MM3DGS_NO_FUSED_PROJECT=1  mm3dgs_slam_map: backward projection and next projection as two launches                       (bit-identical)
 *   MM3DGS_NO_POSE_CHAIN=1     mm3dgs_slam_track: gradient records + backward projection instead of the compositor's pose chain (ABI 208)
 *   MM3DGS_NO_TILE_ORDER=1     arithmetic workgroup -> tile map instead of the load-balanced table                           (bit-identical)
 *   MM3DGS_TILEMAP=0           workgroup b composites tile b (default 1: a contiguous span of tiles per XCD)                 (bit-identical)
 *   MM3DGS_DIRECT_MAX_TILES=n  largest tile grid that may use direct bins (default 11264)
 *   MM3DGS_STATS=1             count compositor wave steps into Mm3dgsHeader (adds atomics: not for timing)
 *   MM3DGS_SLAM_LDS_PAD / MM3DGS_FWD_LDS_PAD / MM3DGS_BWD_LDS_PAD=bytes   never-touched dynamic LDS per workgroup (occupancy experiments)
 * Read by the Python package, not the library: MM3DGS_LIB=<path> loads a variant build of the library (tools/build_variant.sh).
 * MM3DGS_EXP (timing probes with a phase removed: INVALID results) exists only in a library built with -DMM3DGS_PROBES. */
const char* mm3dgs_last_error(void);
/* ABI version of the library = the version of this header (tests/test_cabi.py holds the two together).
   100: round 1 | 200: double optimiser hyper-parameters, flags in mm3dgs_slam_backward, direct bins, map surgery entry points
   201: Mm3dgsLossConfig grew by the three splatam fields | 202: mm3dgs_propagate_const_vel, per-tile gradient records (binning / scratch sizes grew)
   203: Mm3dgsMapView.dpose_out_or_null, header.tile_order_tiles = image-size key, overflowing iterations are void
   204: mm3dgs_slam_adam_project, MM3DGS_FWD_PROJECTED / MM3DGS_FWD_KEEP_TILE_ORDER
   205: Mm3dgsHeader.overflow_seen (appended) | 206: Mm3dgsPoseAdam.best (appended) | 207: Mm3dgsHeader.mean_wave_steps (appended)
   208: mm3dgs_geom_bytes grew by the per-Gaussian pose-chain record (80 B; mm3dgs_slam_track's compositor applies it and writes no gradient
        records); the SLAM modes' block records are addressed by list position and the sorted bin (mask | per-tile record) overwrites the keys:
        binning_state / backward_scratch keep their sizes, their interior layout is the library's own; image_state must be zero-initialised
        to at least sizeof(Mm3dgsHeader) before its first use with MM3DGS_FWD_STATE_CLEAN
   209: Mm3dgsSlamInputs.f_rest / sh_degree / n_rest, Mm3dgsSlamGrads.d_f_rest, Mm3dgsMapAdam.rest_* (all appended): native loops at an active SH degree > 0 */
#define MM3DGS_ABI_VERSION 209
int mm3dgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MM3DGS_H */
