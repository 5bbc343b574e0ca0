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
#include "loss_pixel.h"

#define LT 16
#define HALO 5
#define LW (LT + 2 * HALO)  // 26
#define LWP 27              // LDS row stride (odd: rows start on different banks; 48 would be conflict-free but costs a workgroup of occupancy)
#define NSUM 16

__global__ void __launch_bounds__(256)
loss_reduce_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                   float* __restrict__ dmaps, double* __restrict__ partial) {
  // One colour channel at a time through 14 KB of LDS: the whole grid (1200 workgroups at 640x480) is then resident at
  // once (>= 5 workgroups per CU); staging all three channels (42 KB) left room for 3 per CU -> two rounds, the second
  // nearly empty (24 us instead of ~12).  The next channel's halo loads are in flight while this one is convolved.
  __shared__ float sI[LW][LWP], sG[LW][LWP];
  __shared__ float hM1[LW][LT], hM2[LW][LT], hE11[LW][LT], hE22[LW][LT], hE12[LW][LT];
  __shared__ double red[4][12];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
  const int px = x0 + tx, py = y0 + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const size_t HW = (size_t)cfg.H * cfg.W, pix = (size_t)py * cfg.W + px;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;

  const float sil = inside ? out[4 * HW + pix] : 0.f;
  if (cfg.w_ssim != 0.f) {
    // the (row, column) of a lane's three halo elements are computed once and reused for every channel
    constexpr int NEL = (LW * LW + 255) / 256;   // 3
    int off[NEL], lds[NEL];
#pragma unroll
    for (int e = 0; e < NEL; e++) {
      const int i = threadIdx.x + e * 256;
      const int ly = i / LW, lx = i - ly * LW;
      const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
      const bool in = i < LW * LW && gx >= 0 && gx < cfg.W && gy >= 0 && gy < cfg.H;
      off[e] = in ? gy * cfg.W + gx : -1;
      lds[e] = i < LW * LW ? ly * LWP + lx : -1;
    }
    float va[NEL], vb[NEL];
#pragma unroll
    for (int e = 0; e < NEL; e++) {
      va[e] = off[e] >= 0 ? out[off[e]] : 0.f;
      vb[e] = off[e] >= 0 ? gt[off[e]] : 0.f;
    }
    for (int ch = 0; ch < 3; ch++) {
#pragma unroll
      for (int e = 0; e < NEL; e++)
        if (lds[e] >= 0) { (&sI[0][0])[lds[e]] = va[e]; (&sG[0][0])[lds[e]] = vb[e]; }
      __syncthreads();   // also orders the previous channel's vertical pass before this channel's h* writes
      if (ch < 2) {
#pragma unroll
        for (int e = 0; e < NEL; e++) {
          va[e] = off[e] >= 0 ? out[(size_t)(ch + 1) * HW + off[e]] : 0.f;
          vb[e] = off[e] >= 0 ? gt[(size_t)(ch + 1) * HW + off[e]] : 0.f;
        }
      }
      // horizontal pass: 26 rows x 16 columns = 416 outputs, lane t takes outputs t and t + 256 (shift/mask only)
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int o = threadIdx.x + e * 256;
        if (o < LW * LT) {
          const int ly = o >> 4, lx = o & 15;
          const float* ri = &sI[ly][lx];
          const float* rg = &sG[ly][lx];
          float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
          for (int k = 0; k < 11; k++) {
            const float a = ri[k], b = rg[k], w = cfg.window[k];
            const float wa = w * a, wb = w * b;
            m1 += wa; m2 += wb; e11 = fmaf(wa, a, e11); e22 = fmaf(wb, b, e22); e12 = fmaf(wa, b, e12);
          }
          hM1[ly][lx] = m1; hM2[ly][lx] = m2; hE11[ly][lx] = e11; hE22[ly][lx] = e22; hE12[ly][lx] = e12;
        }
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
        dmaps[(size_t)(ch * 3 + 0) * HW + pix] = df_dm1 - 2.f * m1 * df_ds1 - m2 * df_ds12;  // d/d mu1 (total)
        dmaps[(size_t)(ch * 3 + 1) * HW + pix] = df_ds1;                                     // d/d E[x^2]
        dmaps[(size_t)(ch * 3 + 2) * HW + pix] = df_ds12;                                    // d/d E[xy]
      }
    }
  }
  if (inside) {
    const float rgb[3] = {out[pix], out[HW + pix], out[2 * HW + pix]}, g3[3] = {gt[pix], gt[HW + pix], gt[2 * HW + pix]};
    loss_px_sums(cfg, rgb, sil, cfg.w_pearson != 0.f ? out[3 * HW + pix] : 0.f, g3, cfg.w_pearson != 0.f ? ref[pix] : 0.f, acc);   // acc[2] (SSIM) untouched
  }
  block_sums<12>(acc, red, cfg.w_pearson != 0.f);
  // one row of partial sums per workgroup (plain stores): 14k double atomics on two cache lines cost ~90 us
  if (threadIdx.x == 0) {
    double* row = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) row[k] = acc[k];
  }
}

__device__ __forceinline__ void pearson_terms(double n, double sx, double sxx, double st, double stt, double sxt, double& rho,
                                              double& cxx, double& ctt) {
  cxx = sxx - sx * sx / n;
  ctt = stt - st * st / n;
  const double cxt = sxt - sx * st / n;
  rho = cxt / sqrt(cxx * ctt);
}

// one 1024-lane workgroup: lane = 16 * rowgroup + column; 64 row groups keep the dependent-load chains short (the
// kernel is pure latency), then the groups are added in a fixed order (deterministic).  Lane 0 also derives the Pearson
// scalars every pixel of loss_grad_kernel needs (sums[16..23]) -- ~200 double-precision operations that each of the
// 307 k lanes of that kernel used to repeat.
__global__ void __launch_bounds__(1024) loss_finish_kernel(const double* __restrict__ partial, int nrows, double* __restrict__ sums,
                                                           int pearson_on, int pearson_invert) {
  __shared__ double part[64][16];
  __shared__ double part8[8][16];
  __shared__ double tot[16];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  double a0 = 0.0, a1 = 0.0;
  if (col < 12) {
    int r = grp;
    for (; r + 64 < nrows; r += 128) { a0 += partial[(size_t)r * 12 + col]; a1 += partial[(size_t)(r + 64) * 12 + col]; }
    if (r < nrows) a0 += partial[(size_t)r * 12 + col];
  }
  part[grp][col] = a0 + a1;
  __syncthreads();
  // 64 row groups -> 8 -> 1 in a fixed order (a 64-step serial sum of dependent LDS reads cost ~2 us of pure latency)
  if (grp < 8) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) t += part[grp * 8 + q][col];
    part8[grp][col] = t;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) t += part8[q][threadIdx.x];
    sums[threadIdx.x] = t;
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double valid = 0.0, use2 = 0.0, rho = 0.0, k1 = 0.0, k2 = 0.0, mx = 0.0, mt = 0.0, loss_p = 0.0;
    const double n = tot[3];
    if (pearson_on && n > 1.0) {
      double rho1, cxx, ctt1, rho2 = -2.0, ctt2 = 1.0, cxx2;
      pearson_terms(n, tot[4], tot[5], tot[6], tot[7], tot[8], rho1, cxx, ctt1);
      bool u2 = false;
      if (pearson_invert) {
        pearson_terms(n, tot[4], tot[5], tot[9], tot[10], tot[11], rho2, cxx2, ctt2);
        u2 = (1.0 - rho2) < (1.0 - rho1);
      }
      rho = u2 ? rho2 : rho1;
      const double ctt = u2 ? ctt2 : ctt1;
      valid = 1.0; use2 = u2 ? 1.0 : 0.0;
      k1 = 1.0 / sqrt(cxx * ctt); k2 = rho / cxx;
      mx = tot[4] / n; mt = (u2 ? tot[9] : tot[6]) / n;
      loss_p = 1.0 - rho;
    }
    sums[16] = valid; sums[17] = use2; sums[18] = rho; sums[19] = k1; sums[20] = k2; sums[21] = mx; sums[22] = mt; sums[23] = loss_p;
  }
}


__global__ void __launch_bounds__(256, 5)   // >= 5 waves per SIMD: the whole 1200-workgroup grid resident in one round
loss_grad_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                 const float* __restrict__ dmaps, const double* __restrict__ sums, float* __restrict__ dL, float* __restrict__ loss) {
  __shared__ float sD[3][LW][LWP];   // one colour channel (three derivative maps) at a time, as in loss_reduce_kernel
  __shared__ float hD[3][LW][LT];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
  const int px = x0 + tx, py = y0 + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const size_t HW = (size_t)cfg.H * cfg.W, pix = (size_t)py * cfg.W + px;
  const float l1_scale = loss_l1_scale(cfg, sums);
  const float ssim_scale = -cfg.w_ssim / (float)(3.0 * (double)HW);
  const float sil = inside ? out[4 * HW + pix] : 0.f;
  const bool smask = sil > cfg.sil_thr;
  float gch[3] = {0.f, 0.f, 0.f};
  if (cfg.w_ssim != 0.f) {
    constexpr int NEL = (LW * LW + 255) / 256;   // 3 halo elements per lane and map
    int off[NEL], lds[NEL];
#pragma unroll
    for (int e = 0; e < NEL; e++) {
      const int i = threadIdx.x + e * 256;
      const int ly = i / LW, lx = i - ly * LW;
      const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
      const bool in = i < LW * LW && gx >= 0 && gx < cfg.W && gy >= 0 && gy < cfg.H;
      off[e] = in ? gy * cfg.W + gx : -1;
      lds[e] = i < LW * LW ? ly * LWP + lx : -1;
    }
    float v[3][NEL];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int e = 0; e < NEL; e++) v[q][e] = off[e] >= 0 ? dmaps[(size_t)q * HW + off[e]] : 0.f;
    for (int ch = 0; ch < 3; ch++) {
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < NEL; e++)
          if (lds[e] >= 0) (&sD[q][0][0])[lds[e]] = v[q][e];
      __syncthreads();   // also orders the previous channel's vertical pass before this channel's hD writes
      const float oc = inside ? out[(size_t)ch * HW + pix] : 0.f, gc = inside ? gt[(size_t)ch * HW + pix] : 0.f;
      if (ch < 2) {
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
          for (int e = 0; e < NEL; e++) v[q][e] = off[e] >= 0 ? dmaps[(size_t)((ch + 1) * 3 + q) * HW + off[e]] : 0.f;
      }
#pragma unroll 1
      for (int q = 0; q < 3; q++) {   // not unrolled: 66 LDS reads in flight would cost the fifth wave per SIMD
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int o = threadIdx.x + e * 256;
          if (o < LW * LT) {
            const int ly = o >> 4, lx = o & 15;
            const float* rd = &sD[q][ly][lx];
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) sacc = fmaf(cfg.window[k], rd[k], sacc);
            hD[q][ly][lx] = sacc;
          }
        }
      }
      __syncthreads();
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = cfg.window[k];
        c0 += w * hD[0][ty + k][tx]; c1 += w * hD[1][ty + k][tx]; c2 += w * hD[2][ty + k][tx];
      }
      const float gval = ssim_scale * (c0 + 2.f * oc * c1 + gc * c2);
      if (ch == 0) gch[0] = gval; else if (ch == 1) gch[1] = gval; else gch[2] = gval;
    }
  }
  if (inside) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
      dL[ch * HW + pix] = gch[ch] + loss_px_l1_grad(cfg, out[ch * HW + pix], gt[ch * HW + pix], smask, l1_scale);
    // depth channel: Pearson; silhouette and depth^2 carry no loss
    dL[3 * HW + pix] = cfg.w_pearson != 0.f ? loss_px_pearson_grad(cfg, sil, out[3 * HW + pix], ref[pix], sums) : 0.f;
    dL[4 * HW + pix] = 0.f;
    dL[5 * HW + pix] = 0.f;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && loss) loss_scalars(cfg, sums, HW, loss);
}

void launch_loss_finish(const LossCfg& cfg, double* sums, const double* partial, hipStream_t s) {
  const int nrows = ((cfg.W + LT - 1) / LT) * ((cfg.H + LT - 1) / LT);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, s, partial, nrows, sums, cfg.w_pearson != 0.f ? 1 : 0, cfg.pearson_invert);
}

void launch_loss(const LossCfg& cfg, const float* out, const float* gt, const float* ref, float* dmaps, double* sums, double* partial,
                 float* dL, float* loss, hipStream_t s) {
  dim3 grid((cfg.W + LT - 1) / LT, (cfg.H + LT - 1) / LT), block(256);
  hipLaunchKernelGGL(loss_reduce_kernel, grid, block, 0, s, cfg, out, gt, ref, dmaps, partial);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, s, partial, (int)(grid.x * grid.y), sums, cfg.w_pearson != 0.f ? 1 : 0,
                     cfg.pearson_invert);
  hipLaunchKernelGGL(loss_grad_kernel, grid, block, 0, s, cfg, out, gt, ref, dmaps, sums, dL, loss);
}
