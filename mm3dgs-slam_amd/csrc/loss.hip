// Fused image losses of the SLAM loops, producing dL/d(rendered 6-channel image) directly:
//   tracking  (slam/tracker.py:104-155)  mean |image - gt| over { silhouette > 0.99 } [+ w * Pearson(depth, ref)]
//   mapping   (slam/mapper.py:856-873, utils/loss_utils.py:64-68,95-154)
//             (1-l) * mean |image - gt| + l * (1 - SSIM_11x11,sigma1.5(image, gt)) [+ w * (1 - Pearson(depth, ref))]
// replacing ~60 small torch kernels, 10 MIOpen convolutions and the boolean-mask gathers (host syncs) per iteration.
//
// Two launches: (1) per 16x16 tile, separable Gaussian moments in LDS -> SSIM value and its partial-derivative maps,
// plus every scalar reduction (L1 sum/count, SSIM sum, Pearson moments) accumulated in double; (2) per tile, the
// adjoint convolution of the derivative maps + L1 sign + Pearson gradient -> dL/dout, and the loss scalars.
// Channel layout of `out`: 0..2 RGB, 3 depth (alpha-weighted z), 4 silhouette, 5 depth^2.
#include "mm3dgs_common.h"
#include "fused_api.h"

#define LT 16
#define HALO 5
#define LW (LT + 2 * HALO)  // 26
#define NSUM 16

__device__ __forceinline__ double block_sum(double v, double* sh) {
  // 256 lanes -> lane 0 (wave shuffle then LDS across the 4 waves)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void __launch_bounds__(256)
loss_reduce_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                   float* __restrict__ dmaps, double* __restrict__ sums) {
  __shared__ float sI[LW][LW + 1], sG[LW][LW + 1];
  __shared__ float hM1[LW][LT], hM2[LW][LT], hE11[LW][LT], hE22[LW][LT], hE12[LW][LT];
  __shared__ double red[4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
  const int px = x0 + tx, py = y0 + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const size_t HW = (size_t)cfg.H * cfg.W, pix = (size_t)py * cfg.W + px;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  double acc[NSUM];
#pragma unroll
  for (int k = 0; k < NSUM; k++) acc[k] = 0.0;

  const float sil = inside ? out[4 * HW + pix] : 0.f;
  const bool smask = sil > cfg.sil_thr;
  float l1 = 0.f;
  for (int ch = 0; ch < 3; ch++) {
    if (cfg.w_ssim != 0.f) {
      __syncthreads();
      for (int i = threadIdx.x; i < LW * LW; i += 256) {
        const int ly = i / LW, lx = i % LW;
        const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
        const bool in = gx >= 0 && gx < cfg.W && gy >= 0 && gy < cfg.H;
        sI[ly][lx] = in ? out[ch * HW + (size_t)gy * cfg.W + gx] : 0.f;
        sG[ly][lx] = in ? gt[ch * HW + (size_t)gy * cfg.W + gx] : 0.f;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < LW * LT; i += 256) {
        const int ly = i / LT, lx = i % LT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float a = sI[ly][lx + k], b = sG[ly][lx + k], w = cfg.window[k];
          m1 += w * a; m2 += w * b; e11 += w * a * a; e22 += w * b * b; e12 += w * a * b;
        }
        hM1[ly][lx] = m1; hM2[ly][lx] = m2; hE11[ly][lx] = e11; hE22[ly][lx] = e22; hE12[ly][lx] = e12;
      }
      __syncthreads();
      float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = cfg.window[k];
        m1 += w * hM1[ty + k][tx]; m2 += w * hM2[ty + k][tx]; e11 += w * hE11[ty + k][tx];
        e22 += w * hE22[ty + k][tx]; e12 += w * hE12[ty + k][tx];
      }
      if (inside) {
        const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
        const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s1 + s2 + C2;
        const float inv = 1.f / (B1 * B2);
        const float f = A1 * A2 * inv;
        acc[2] += (double)f;
        const float df_dm1 = 2.f * m2 * A2 * inv - f * 2.f * m1 / B1;
        const float df_ds1 = -f / B2;
        const float df_ds12 = 2.f * A1 * inv;
        dmaps[(ch * 3 + 0) * HW + pix] = df_dm1 - 2.f * m1 * df_ds1 - m2 * df_ds12;  // d/d mu1 (total)
        dmaps[(ch * 3 + 1) * HW + pix] = df_ds1;                                     // d/d E[x^2]
        dmaps[(ch * 3 + 2) * HW + pix] = df_ds12;                                    // d/d E[xy]
      }
    }
    if (inside) l1 += fabsf(out[ch * HW + pix] - gt[ch * HW + pix]);
  }
  if (inside && (cfg.l1_mask == 0 || smask)) { acc[0] = (double)l1; acc[1] = 1.0; }
  if (cfg.w_pearson != 0.f && inside) {
    const float r = ref[pix];
    bool m = true;
    if (cfg.pearson_mask & 1) m = m && smask;
    if (cfg.pearson_mask & 2) m = m && (r > 0.f);
    if (m) {
      const double x = (double)out[3 * HW + pix];
      const double t1 = cfg.pearson_invert ? -(double)r : (double)r;
      const double t2 = 1.0 / ((double)r + 200.0);
      acc[3] = 1.0; acc[4] = x; acc[5] = x * x;
      acc[6] = t1; acc[7] = t1 * t1; acc[8] = x * t1;
      acc[9] = t2; acc[10] = t2 * t2; acc[11] = x * t2;
    }
  }
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const double s = block_sum(acc[k], red);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(&sums[k], s);
  }
}

__device__ __forceinline__ void pearson_terms(double n, double sx, double sxx, double st, double stt, double sxt, double& rho,
                                              double& cxx, double& ctt) {
  cxx = sxx - sx * sx / n;
  ctt = stt - st * st / n;
  const double cxt = sxt - sx * st / n;
  rho = cxt / sqrt(cxx * ctt);
}

__global__ void __launch_bounds__(256)
loss_grad_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                 const float* __restrict__ dmaps, const double* __restrict__ sums, float* __restrict__ dL, float* __restrict__ loss) {
  __shared__ float sD[3][LW][LW + 1];
  __shared__ float hD[3][LW][LT];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
  const int px = x0 + tx, py = y0 + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const size_t HW = (size_t)cfg.H * cfg.W, pix = (size_t)py * cfg.W + px;
  const double n_l1 = sums[1];
  const float l1_scale = n_l1 > 0.0 ? cfg.w_l1 / (float)(3.0 * n_l1) : 0.f;
  const float ssim_scale = -cfg.w_ssim / (float)(3.0 * (double)HW);
  const float sil = inside ? out[4 * HW + pix] : 0.f;
  const bool smask = sil > cfg.sil_thr;
  for (int ch = 0; ch < 3; ch++) {
    float g = 0.f;
    if (cfg.w_ssim != 0.f) {
      __syncthreads();
      for (int i = threadIdx.x; i < 3 * LW * LW; i += 256) {
        const int q = i / (LW * LW), r = i % (LW * LW), ly = r / LW, lx = r % LW;
        const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
        const bool in = gx >= 0 && gx < cfg.W && gy >= 0 && gy < cfg.H;
        sD[q][ly][lx] = in ? dmaps[(ch * 3 + q) * HW + (size_t)gy * cfg.W + gx] : 0.f;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 3 * LW * LT; i += 256) {
        const int q = i / (LW * LT), r = i % (LW * LT), ly = r / LT, lx = r % LT;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) s += cfg.window[k] * sD[q][ly][lx + k];
        hD[q][ly][lx] = s;
      }
      __syncthreads();
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = cfg.window[k];
        c0 += w * hD[0][ty + k][tx]; c1 += w * hD[1][ty + k][tx]; c2 += w * hD[2][ty + k][tx];
      }
      if (inside) g = ssim_scale * (c0 + 2.f * out[ch * HW + pix] * c1 + gt[ch * HW + pix] * c2);
    }
    if (inside) {
      if (cfg.l1_mask == 0 || smask) {
        const float d = out[ch * HW + pix] - gt[ch * HW + pix];
        g += l1_scale * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      }
      dL[ch * HW + pix] = g;
    }
  }
  // depth channel: Pearson
  float gd = 0.f;
  double loss_p = 0.0;
  if (cfg.w_pearson != 0.f) {
    const double n = sums[3];
    if (n > 1.0) {
      double rho1, cxx, ctt1, rho2 = -2.0, ctt2 = 1.0, cxx2;
      pearson_terms(n, sums[4], sums[5], sums[6], sums[7], sums[8], rho1, cxx, ctt1);
      bool use2 = false;
      if (cfg.pearson_invert) {
        pearson_terms(n, sums[4], sums[5], sums[9], sums[10], sums[11], rho2, cxx2, ctt2);
        use2 = (1.0 - rho2) < (1.0 - rho1);
      }
      const double rho = use2 ? rho2 : rho1, ctt = use2 ? ctt2 : ctt1;
      loss_p = 1.0 - rho;
      if (inside) {
        const float r = ref[pix];
        bool m = true;
        if (cfg.pearson_mask & 1) m = m && smask;
        if (cfg.pearson_mask & 2) m = m && (r > 0.f);
        if (m) {
          const double x = (double)out[3 * HW + pix];
          const double t = use2 ? 1.0 / ((double)r + 200.0) : (cfg.pearson_invert ? -(double)r : (double)r);
          const double st = use2 ? sums[9] : sums[6];
          const double drho = (t - st / n) / sqrt(cxx * ctt) - rho * (x - sums[4] / n) / cxx;
          gd = (float)(-(double)cfg.w_pearson * drho);
        }
      }
    }
  }
  if (inside) {
    dL[3 * HW + pix] = gd;
    dL[4 * HW + pix] = 0.f;
    dL[5 * HW + pix] = 0.f;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && loss) {
    const double l1 = n_l1 > 0.0 ? sums[0] / (3.0 * n_l1) : 0.0;
    const double ss = cfg.w_ssim != 0.f ? 1.0 - sums[2] / (3.0 * (double)HW) : 0.0;
    loss[1] = (float)l1; loss[2] = (float)ss; loss[3] = (float)loss_p;
    loss[0] = (float)(cfg.w_l1 * l1 + cfg.w_ssim * ss + cfg.w_pearson * loss_p);
  }
}

void launch_loss(const LossCfg& cfg, const float* out, const float* gt, const float* ref, float* dmaps, double* sums, float* dL,
                 float* loss, hipStream_t s) {
  dim3 grid((cfg.W + LT - 1) / LT, (cfg.H + LT - 1) / LT), block(256);
  (void)hipMemsetAsync(sums, 0, NSUM * sizeof(double), s);
  hipLaunchKernelGGL(loss_reduce_kernel, grid, block, 0, s, cfg, out, gt, ref, dmaps, sums);
  hipLaunchKernelGGL(loss_grad_kernel, grid, block, 0, s, cfg, out, gt, ref, dmaps, sums, dL, loss);
}
