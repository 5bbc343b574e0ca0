// Per-pixel pieces of the SLAM image losses that need no neighbourhood (masked L1, Pearson moments and their gradient),
// shared by the loss kernels (loss.hip) and by the compositors when a tracking iteration folds the loss into them
// (composite.hip): one source for both.  Channel layout: rgb[3], depth (alpha-weighted z), sil (accumulated alpha).
// Reference: slam/tracker.py:104-155, slam/mapper.py:856-873, utils/loss_utils.py:43-61.
#pragma once
#include "mm3dgs_common.h"
#include "fused_api.h"

// sum of the twelve per-lane loss values over the 256-lane workgroup (valid in lane 0 only).  Columns 0..3 (L1 sum / count, SSIM
// sum, Pearson pixel count: sums of non-negative terms, <= 64 addends per wave) take a float DPP reduction; the Pearson moments
// x, xx, t, tt, xt (columns 4..11; the covariances are later formed as stt - st^2 / n, where mean^2 / variance reaches 1e5 on the
// 1 / (ref + 200) branch, so a 1e-7 relative rounding of the sums would become 1e-2 on the correlation) stay in double -- only
// the columns the configuration uses (workgroup-uniform mask): none without a Pearson term, 4..8 for the mapping loss, 4..11 with
// the two-target tracking form.
__device__ __forceinline__ unsigned pearson_double_cols(const LossCfg& cfg) {
  return cfg.w_pearson == 0.f ? 0u : (cfg.pearson_invert ? 0xff0u : 0x1f0u);
}
template <int NR>
__device__ __forceinline__ void block_sums(double (&v)[NR], double (*sh)[NR], unsigned dmask) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NR; k++) {
    if (k < 4) {
      const float t = wave_sum_to_lane63((float)v[k]);
      if (lane == 63) sh[wv][k] = (double)t;
    } else if ((dmask >> k) & 1u) {
      const double t = wave_sum_to_lane63_f64(v[k]);
      if (lane == 63) sh[wv][k] = t;
    } else if (lane == 63) {
      sh[wv][k] = 0.0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < NR; k++) v[k] = sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k];
}

// this pixel's contribution to the twelve loss sums: [0] L1 sum, [1] L1 count, [2] SSIM sum (not touched here),
// [3..11] Pearson moments n, x, xx, t1, t1t1, xt1, t2, t2t2, xt2  (x = rendered depth, t1 = +-ref, t2 = 1/(ref+200))
__device__ __forceinline__ void loss_px_sums(const LossCfg& cfg, const float (&rgb)[3], float sil, float depth, const float (&gt)[3], float refv,
                                             double (&acc)[12]) {
  const bool smask = sil > cfg.sil_thr;
  if (cfg.l1_mask == 0 || smask) {
    float l1 = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) l1 += fabsf(rgb[ch] - gt[ch]);
    acc[0] = (double)l1;
    acc[1] = 1.0;
  }
  if (cfg.w_pearson != 0.f) {
    bool m = true;
    if (cfg.pearson_mask & 1) m = m && smask;
    if (cfg.pearson_mask & 2) m = m && (refv > 0.f);
    if (m) {
      const double x = (double)depth;
      const double t1 = cfg.pearson_invert ? -(double)refv : (double)refv;
      const double t2 = 1.0 / ((double)refv + 200.0);
      acc[3] = 1.0; acc[4] = x; acc[5] = x * x;
      acc[6] = t1; acc[7] = t1 * t1; acc[8] = x * t1;
      acc[9] = t2; acc[10] = t2 * t2; acc[11] = x * t2;
    }
  }
}

// L1 gradient of one colour channel (sums[1] = number of pixels in the L1 mean)
__device__ __forceinline__ float loss_px_l1_grad(const LossCfg& cfg, float v, float g, bool smask, float l1_scale) {
  if (!(cfg.l1_mask == 0 || smask)) return 0.f;
  const float d = v - g;
  return l1_scale * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
}
__device__ __forceinline__ float loss_l1_scale(const LossCfg& cfg, const double* sums) {
  const double n_l1 = sums[1];
  return n_l1 > 0.0 ? cfg.w_l1 / (float)(3.0 * n_l1) : 0.f;
}

// Pearson gradient w.r.t. the rendered depth: d(1 - rho)/dx = -[(t - mean_t) / sqrt(cxx ctt) - rho (x - mean_x) / cxx], with
// the scalars loss_finish_kernel prepared in sums[16..23]
__device__ __forceinline__ float loss_px_pearson_grad(const LossCfg& cfg, float sil, float depth, float refv, const double* sums) {
  if (cfg.w_pearson == 0.f || sums[16] == 0.0) return 0.f;
  const bool smask = sil > cfg.sil_thr;
  bool m = true;
  if (cfg.pearson_mask & 1) m = m && smask;
  if (cfg.pearson_mask & 2) m = m && (refv > 0.f);
  if (!m) return 0.f;
  const bool use2 = sums[17] != 0.0;
  const double x = (double)depth;
  const double t = use2 ? 1.0 / ((double)refv + 200.0) : (cfg.pearson_invert ? -(double)refv : (double)refv);
  const double drho = (t - sums[22]) * sums[19] - sums[20] * (x - sums[21]);
  return (float)(-(double)cfg.w_pearson * drho);
}

// the four loss scalars {total, l1, 1-ssim, 1-rho} from the finished sums
__device__ __forceinline__ void loss_scalars(const LossCfg& cfg, const double* sums, size_t HW, float* loss) {
  const double n_l1 = sums[1];
  const double l1 = n_l1 > 0.0 ? sums[0] / (3.0 * n_l1) : 0.0;
  const double ss = cfg.w_ssim != 0.f ? 1.0 - sums[2] / (3.0 * (double)HW) : 0.0;
  const double loss_p = (cfg.w_pearson != 0.f && sums[16] != 0.0) ? sums[23] : 0.0;
  loss[1] = (float)l1; loss[2] = (float)ss; loss[3] = (float)loss_p;
  loss[0] = (float)(cfg.w_l1 * l1 + cfg.w_ssim * ss + cfg.w_pearson * loss_p);
}
