// Internal structs + launchers of the fused SLAM path (fused.hip, loss.hip), shared with api.hip.
#pragma once
#include "mm3dgs_common.h"

struct SlamIn {
  const float* pose; const float* xyz; const float* f_dc; const float* opacity; const float* scaling; const float* rotation;
  int isotropic;
  int world;     // Mm3dgsSlamInputs.world_means
  const float* f_rest; int sh_deg; int n_rest;      // active SH degree > 0 (ABI 209): rows [P][n_rest][3]; sh_deg == 0: unused
};
struct SlamGrads {
  float* d_xyz; float* d_f_dc; float* d_opacity; float* d_scaling; float* d_rotation;
  float* max_radii2D; float* grad_accum; float* denom;
  float* d_f_rest;
};
// Adam scalars as torch.optim.Adam applies them: formed in double on the host, rounded once to float.
//   omb1 = 1 - beta1 (lerp weight), beta2, omb2 = 1 - beta2, step_size = lr / (1 - beta1^t) per group, bc2s = sqrt(1 - beta2^t)
struct MapAdam { float* p[5]; float* m[5]; float* v[5]; float step_size[5]; float omb1, beta2, omb2, eps, bc2s; int on; const uint8_t* opt_mask;
                 float* rp; float* rm; float* rv; float rest_step_size; };      // (the sixth group: f_rest, active SH degree > 0)
struct PoseAdam { float* pose; float* m; float* v; int* step; double lr_q, lr_t, beta1, beta2; float eps; const float* prior; float prior_w_t, prior_w_q; float* best; };
struct AdamGroup { float* p; const float* g; float* m; float* v; unsigned long long n; float step_size; };
struct AdamArgs { AdamGroup grp[8]; int ngroups; float omb1, beta2, omb2, eps, bc2s; };
struct LossCfg {
  int H, W;
  float w_l1, w_ssim, w_pearson;
  int l1_mask, pearson_mask, pearson_invert;
  float sil_thr;
  float window[11];
  // `method: splatam` forms (standalone loss kernels only; api.hip never folds such a configuration into the compositors):
  float w_depth;       // weight of the depth-L1 term |ref - depth| over depth_mask (bit0 silhouette > sil_thr, bit1 ref > 0)
  int depth_mask;
  int l1_sum;          // the L1 terms are sums over their masks instead of means;  l1_mask bit1: ref > 0
};

// A tracking iteration without SSIM folds the per-pixel loss into the compositors (mm3dgs_slam_track): the forward epilogue
// writes the per-tile partial sums loss_reduce_kernel would, the backward prologue derives dL/d(image) from the finished
// sums instead of reading the gradient image loss_grad_kernel would have written.  Two launches and a 7 MB round trip less.
struct TrackLoss {
  LossCfg cfg;
  const float* gt;      // [3,H,W]
  const float* ref;     // [H,W] or NULL (Pearson off)
  const float* out;     // [6,H,W] the rendered bundle (backward prologue)
  const float* dmaps;   // [9,H,W] SSIM derivative maps (mapping loss folded into the backward compositor's prologue) or NULL
  double* partial;      // [tiles][12]
  const double* sums;   // finished sums (loss_finish_kernel)
  float* loss4;         // {total, l1, 1-ssim, 1-rho} or NULL
  int defer_scale;      // 1: masked-L1 only -- the 1/(3 n) normalisation is applied to the pose gradient by the pose
                        //    finishing kernel (the gradient is linear in dL), so no loss-finish launch is needed at all
};
// the loss partial rows the pose finishing kernel sums when the normalisation is deferred (see TrackLoss::defer_scale)
struct PoseLossScale { const double* rows; int nrows; float w_l1; float* loss4; };
void launch_loss_finish(const LossCfg& cfg, double* sums, const double* partial, hipStream_t s, float* loss4 = nullptr);

void launch_slam_preprocess_fwd(const CamDev& cam, int P, const SlamIn& in, int32_t* radii, GeomView g, ImageView iv, hipStream_t s,
                                uint32_t* seen_only = nullptr, bool visibility_only = false, bool want_poserec = false);
bool composite_has_pose_chain();
// the pose finish alone over per-tile pose rows (the tracking compositor's pose chain: composite.hip, GeomView.poserec)
void launch_slam_pose_finish(const float* rows, int nrows, const float* pose_in, float* dpose, const PoseAdam& ad, hipStream_t s,
                             const PoseLossScale* pls, float* loss4, const uint32_t* ovf);
void launch_slam_preprocess_bwd(const CamDev& cam, int P, const SlamIn& in, const int32_t* radii, GeomView g, BinView b, size_t N_cap,
                                BwdView bw, const SlamGrads& out, float* dpose, const PoseAdam& ad, const MapAdam& ma, hipStream_t s,
                                const PoseLossScale* pls = nullptr, float* loss4 = nullptr, bool direct = false, const uint32_t* overflow_flag = nullptr);
// dl_planes: 6, or 4 when the caller guarantees that the silhouette / depth^2 planes of dL are zero AND need not be read
// (the mapping loop's loss kernel does not even write them)
// pose_chain (tracking only): apply GeomView.poserec per (block, splat) and write the tile's pose-gradient row to dsub[tile][32] instead of
// gradient records
void launch_composite_bwd_slam(const CamDev& cam, bool tracking, GeomView g, ImageView iv, BinView b, size_t N_cap, const float* dL,
                               float* dsub, hipStream_t s, const TrackLoss* tl = nullptr, int dl_planes = 6, bool pose_chain = false);
// sort + forward compositing of the 6-channel SLAM bundle in one launch (lists <= 2048 per tile stay in LDS)
void launch_sort_composite_fwd6(const CamDev& cam, GeomView g, ImageView iv, BinView b, size_t N_cap, float* out, int clean, hipStream_t s,
                                const TrackLoss* tl = nullptr, int direct_blocks = 0, uint32_t direct_cap = 0, int slot_bits = DIRECT_SLOT_BITS_MAX);
// the same + the backward compositor of a tracking iteration (masked-L1 loss, deferred normalisation) in that launch
void launch_sort_composite_fwd_bwd_track(const CamDev& cam, GeomView g, ImageView iv, BinView b, size_t N_cap, float* out, int clean,
                                         hipStream_t s, const TrackLoss& tl, int direct_blocks, float* dsub, uint32_t direct_cap = 0, int slot_bits = DIRECT_SLOT_BITS_MAX,
                                         bool pose_chain = false);
// backward projection + map Adam of one mapping iteration and projection + binning of the next one (its pose: next_pose) in one launch
void launch_slam_bwd_project(const CamDev& cam, int P, const SlamIn& in, int32_t* radii, GeomView g, ImageView iv, BinView b, size_t N_cap,
                             BwdView bw, const SlamGrads& out, const MapAdam& ma, const float* next_pose, uint32_t bin_cap, uint32_t rec_cap,
                             int slot_bits, hipStream_t s);
// projection + binning in one launch (direct bins: every tile owns bin_cap pairs at tile * bin_cap)
void launch_slam_project_bin(const CamDev& cam, int P, const SlamIn& in, int32_t* radii, GeomView g, ImageView iv, BinView b, uint32_t bin_cap,
                             uint32_t rec_cap, int slot_bits, hipStream_t s, bool want_poserec = false);
// the map's Adam step from gradient arrays (the multi-GPU window: all-reduced gradients) + projection + binning of the next view in one launch
void launch_slam_adam_project(const CamDev& cam, int P, const SlamIn& in, int32_t* radii, GeomView g, ImageView iv, BinView b, const SlamGrads& gr,
                              const MapAdam& ma, const float* next_pose, uint32_t bin_cap, uint32_t rec_cap, int slot_bits, hipStream_t s);
void launch_fused_adam(const AdamArgs& a, hipStream_t s);
void launch_loss(const LossCfg& cfg, const float* out, const float* gt, const float* ref, float* dmaps, double* sums, double* partial,
                 float* dL, float* loss, hipStream_t s);
void launch_loss_after_forward_rows(const LossCfg& cfg, const float* out, const float* gt, const float* ref, float* dmaps, double* sums,
                                    double* partial, float* dL, hipStream_t s);

// ---- map surgery (compact.hip) ----
struct CompactTable { const float* src[32]; float* dst[32]; int width[32]; int n_arrays; };
struct SeedOut { float* xyz; float* f_dc; float* f_rest; float* opacity; float* scaling; float* rotation; float* rgb; int n_rest; };
void launch_prune_mask(int P, const float* opacity, const float* scaling, const float* max_radii2D, float min_opacity, float max_scale,
                       float max_screen_size, int use_screen, uint8_t* keep, uint32_t* n_pruned, hipStream_t s);
void launch_compact_plan(int n, const uint8_t* keep, uint32_t* block_pre, uint32_t* n_keep, hipStream_t s);
void launch_compact_rows(int n, const uint8_t* keep, const uint32_t* block_pre, const CompactTable& t, hipStream_t s);
void launch_seed_gaussians(int H, int W, const float* color, const float* depth, const uint8_t* keep, const uint32_t* block_pre,
                           const float* pose, float fx, float fy, float cx, float cy, uint32_t row0, const SeedOut& o, hipStream_t s);

void launch_covisibility_ratio(int H, int W, const float* depth, const float* sil, const float* kf_pose, const float* cur_pose, float fx, float fy,
                               float cx, float cy, uint32_t* counts, hipStream_t s);
void launch_propagate_const_vel(const float* pm1, const float* pm2, float* out, hipStream_t s);
