// Fused image losses of the SLAM loops, producing dL/d(rendered 6-channel image) directly:
//   tracking  (slam/tracker.py:104-155)  mean |image - gt| over { silhouette > 0.99 } [+ w * Pearson(depth, ref)]
//   mapping   (slam/mapper.py:856-873, utils/loss_utils.py:64-68,95-154)
//             (1-l) * mean |image - gt| + l * (1 - SSIM_11x11,sigma1.5(image, gt)) [+ w * (1 - Pearson(depth, ref))]
// replacing ~60 small torch kernels, 10 MIOpen convolutions and the boolean-mask gathers (host syncs) per iteration.
//
// Kernels (16x16-pixel tile per 256-lane workgroup; the whole 1200-workgroup grid of a 640x480 image is resident at once):
//   ssim_maps_kernel   separable 11x11 Gaussian moments of one colour channel at a time -> SSIM value and its three
//                      partial-derivative maps.  Both passes work on register strips: the horizontal pass reads 14 raw values
//                      per image with four ds_read_b128 and produces 4 outputs x 5 moments (220 FMA per 8 LDS reads; the first
//                      version issued one ds_read_b32 per FMA operand and ran at 6 % of the HBM roofline), the vertical pass
//                      produces 2 outputs per lane from 12 rows.  <ROWS>: also the per-pixel sums (L1, Pearson moments) of the
//                      tile; otherwise those rows were written by the forward compositor's epilogue (mm3dgs_slam_map) and one
//                      extra workgroup of this launch reduces them to the scalars the gradient kernel needs -- no separate
//                      finishing launch in the mapping loop.
//   loss_rows_kernel   the per-pixel sums alone (losses without an SSIM term outside the folded tracking path).
//   loss_finish_kernel one workgroup: fixed-order double-precision sum of the tile rows, Pearson scalars, the four loss values.
//   loss_grad_kernel   adjoint convolution of the derivative maps (same strip scheme) + L1 sign + Pearson gradient -> dL/dout.
// Channel layout of `out`: 0..2 RGB, 3 depth (alpha-weighted z), 4 silhouette, 5 depth^2.
#include "loss_tile.h"

// fixed-order reduction of the tile rows by one 256-lane workgroup: lane = 16 * rowgroup + column.  cols_mask selects the
// columns to sum (a column that another workgroup of the same launch is still writing must not be read).
__device__ __forceinline__ void reduce_rows_256(const double* __restrict__ partial, int nrows, unsigned cols_mask, double (*part)[16], double* tot) {
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (col < 12 && ((cols_mask >> col) & 1u)) {
    int r = grp;
    for (; r + 48 < nrows; r += 64) {
      a0 += partial[(size_t)r * 12 + col]; a1 += partial[(size_t)(r + 16) * 12 + col];
      a2 += partial[(size_t)(r + 32) * 12 + col]; a3 += partial[(size_t)(r + 48) * 12 + col];
    }
    for (; r < nrows; r += 16) a0 += partial[(size_t)r * 12 + col];
  }
  part[grp][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) t += part[q][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

__device__ __forceinline__ void pearson_terms(double n, double sx, double sxx, double st, double stt, double sxt, double& rho,
                                              double& cxx, double& ctt) {
  cxx = sxx - sx * sx / n;
  ctt = stt - st * st / n;
  const double cxt = sxt - sx * st / n;
  rho = cxt / sqrt(cxx * ctt);
}

// lane 0: the Pearson scalars every pixel of the gradient pass needs (sums[16..23]) from the finished moments tot[3..11]
__device__ __forceinline__ void pearson_scalars(const double* tot, int pearson_on, int pearson_invert, double* __restrict__ sums) {
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

// ---- `method: splatam` forms of the per-pixel terms (slam/tracker.py:110-126, slam/mapper.py:836-855) ----------------------------
// colour L1 over { silhouette > thr } and / or { ref > 0 }, a depth-L1 term |ref - depth| over its own mask, both as means or as
// sums.  Row columns: [0] colour L1 sum, [1] its pixel count, [3] depth pixel count, [4] depth L1 sum -- the first two Pearson
// columns, which is why the API refuses the two terms together.  Only the standalone kernels of this file know these forms.
__device__ __forceinline__ bool loss_variant(const LossCfg& cfg) { return cfg.w_depth != 0.f || cfg.l1_sum != 0 || (cfg.l1_mask & 2) != 0; }
__device__ __forceinline__ unsigned loss_double_cols(const LossCfg& cfg) { return pearson_double_cols(cfg) | (cfg.w_depth != 0.f ? 0x10u : 0u); }
__device__ __forceinline__ void variant_px_sums(const LossCfg& cfg, float l1, float sil, float depth, float refv, double (&acc)[12]) {
  const bool smask = sil > cfg.sil_thr;
  const bool on = loss_mask_on(cfg.l1_mask, smask, refv);
  acc[0] = on ? (double)l1 : 0.0;
  acc[1] = on ? 1.0 : 0.0;
  if (cfg.w_depth != 0.f) {
    const bool don = loss_mask_on(cfg.depth_mask, smask, refv);
    acc[3] = don ? 1.0 : 0.0;
    acc[4] = don ? fabs((double)refv - (double)depth) : 0.0;
  }
}
// {total, colour l1, 1-ssim, depth term (or 1-rho)} of a configuration with these forms
__device__ __forceinline__ void variant_scalars(const LossCfg& cfg, const double* sums, size_t HW, float* loss) {
  const double n_l1 = sums[1], n_d = sums[3];
  const double l1 = cfg.l1_sum ? sums[0] : (n_l1 > 0.0 ? sums[0] / (3.0 * n_l1) : 0.0);
  const double ss = cfg.w_ssim != 0.f ? 1.0 - sums[2] / (3.0 * (double)HW) : 0.0;
  const double loss_p = (cfg.w_pearson != 0.f && sums[16] != 0.0) ? sums[23] : 0.0;
  const double dl = cfg.w_depth != 0.f ? (cfg.l1_sum ? sums[4] : (n_d > 0.0 ? sums[4] / n_d : 0.0)) : 0.0;
  loss[1] = (float)l1; loss[2] = (float)ss; loss[3] = (float)(cfg.w_depth != 0.f ? dl : loss_p);
  loss[0] = (float)(cfg.w_l1 * l1 + cfg.w_ssim * ss + cfg.w_pearson * loss_p + cfg.w_depth * dl);
}

// ---- SSIM moments and derivative maps ---------------------------------------------------------------------------------------
template <bool ROWS>
__global__ void __launch_bounds__(256, 5)   // >= 5 waves per SIMD: the whole 1200-workgroup grid of a 640x480 image in one round
ssim_maps_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                 float* __restrict__ dmaps, double* __restrict__ partial, double* __restrict__ sums, int tiles_x, int T) {
  __shared__ __align__(16) float sI[LW][SW], sG[LW][SW];
  __shared__ __align__(16) float hS[5][LW][HW_];
  __shared__ double red[4][12];
  if ((int)blockIdx.x >= T) {
    // the extra workgroup of the mapping loop: the forward compositor's epilogue wrote the L1 / Pearson rows of this render;
    // reduce them (every column but the SSIM sums the other workgroups are writing right now) and derive the Pearson scalars
    double (*part)[16] = (double (*)[16])&hS[0][0][0];
    __shared__ double tot[16];
    reduce_rows_256(partial, T, 0xffbu, part, tot);
    if (threadIdx.x < 12 && threadIdx.x != 2) sums[threadIdx.x] = tot[threadIdx.x];
    if (threadIdx.x == 0) pearson_scalars(tot, cfg.w_pearson != 0.f ? 1 : 0, cfg.pearson_invert, sums);
    return;
  }
  const int tile = blockIdx.x;
  const int x0 = (tile % tiles_x) * LT, y0 = (tile / tiles_x) * LT;
  const uint32_t HW = (uint32_t)cfg.H * (uint32_t)cfg.W;   // 32-bit element offsets (9 H W < 2^31 is checked by the API): 64-bit address temporaries spilled
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const int tid = threadIdx.x;
  const HaloIdx hx = halo_index(cfg, x0, y0);
  const Rsrc r_out = make_rsrc(out, HW * 24u), r_gt = make_rsrc(gt, HW * 12u), r_dm = make_rsrc(dmaps, HW * 36u);
  float va[3], vb[3];
#pragma unroll
  for (int e = 0; e < 3; e++) {
    va[e] = hx.off[e] >= 0 ? bld(r_out, (uint32_t)hx.off[e]) : 0.f;
    vb[e] = hx.off[e] >= 0 ? bld(r_gt, (uint32_t)hx.off[e]) : 0.f;
  }
  // per-pixel sums of this lane's pixel (ROWS): lane = 16 * row + column of the tile
  const int tx = tid & 15, ty = tid >> 4;
  const int px = x0 + tx, py = y0 + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const uint32_t pix = (uint32_t)py * (uint32_t)cfg.W + (uint32_t)px;
  float l1 = 0.f;
  float ssim_sum = 0.f;      // 3 addends per lane
#pragma unroll 1   // (unrolled x3 the kernel needed 120+ registers)
  for (int ch = 0; ch < 3; ch++) {
#pragma unroll
    for (int e = 0; e < 3; e++)
      if (hx.lds[e] >= 0) { (&sI[0][0])[hx.lds[e]] = va[e]; (&sG[0][0])[hx.lds[e]] = vb[e]; }
    __syncthreads();   // staging complete; also: every lane has finished the previous channel's vertical pass (hS is free)
    if (ch < 2) {
#pragma unroll
      for (int e = 0; e < 3; e++) {
        va[e] = hx.off[e] >= 0 ? bld(r_out, (uint32_t)(ch + 1) * HW + (uint32_t)hx.off[e]) : 0.f;
        vb[e] = hx.off[e] >= 0 ? bld(r_gt, (uint32_t)(ch + 1) * HW + (uint32_t)hx.off[e]) : 0.f;
      }
    }
    if (ROWS) l1 += fabsf(sI[ty + HALO][tx + HALO] - sG[ty + HALO][tx + HALO]);
    // horizontal pass: 26 rows x 8 strips of 2 outputs (208 of the 256 lanes); 12 inputs per image as six 8-byte reads
    if (tid < LW * 8) {
      const int row = tid >> 3, s = tid & 7;
      float a[12], b[12];
      const float2* ra = (const float2*)&sI[row][2 * s];
      const float2* rb = (const float2*)&sG[row][2 * s];
#pragma unroll
      for (int v = 0; v < 6; v++) {
        const float2 qa = ra[v], qb = rb[v];
        a[2 * v] = qa.x; a[2 * v + 1] = qa.y; b[2 * v] = qb.x; b[2 * v + 1] = qb.y;
      }
      float m1[2] = {0.f, 0.f}, m2[2] = {0.f, 0.f}, e11[2] = {0.f, 0.f}, e22[2] = {0.f, 0.f}, e12[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const float aa = a[i] * a[i], bb = b[i] * b[i], ab = a[i] * b[i];
#pragma unroll
        for (int o = 0; o < 2; o++) {
          const int k = i - o;
          if (k >= 0 && k < 11) {
            const float w = cfg.window[k];
            m1[o] = fmaf(w, a[i], m1[o]); m2[o] = fmaf(w, b[i], m2[o]);
            e11[o] = fmaf(w, aa, e11[o]); e22[o] = fmaf(w, bb, e22[o]); e12[o] = fmaf(w, ab, e12[o]);
          }
        }
      }
      *(float2*)&hS[0][row][2 * s] = make_float2(m1[0], m1[1]);
      *(float2*)&hS[1][row][2 * s] = make_float2(m2[0], m2[1]);
      *(float2*)&hS[2][row][2 * s] = make_float2(e11[0], e11[1]);
      *(float2*)&hS[3][row][2 * s] = make_float2(e22[0], e22[1]);
      *(float2*)&hS[4][row][2 * s] = make_float2(e12[0], e12[1]);
    }
    __syncthreads();
    // vertical pass: one output per lane (all 256), 11 rows of each moment.  (Two horizontally adjacent outputs per lane on 128 lanes -- half the
    // LDS reads -- keep ten moments live in a kernel that is held to 96 registers: 7 - 14 of them spill, 15.8 us instead of 12.3: DESIGN.md section 4)
    constexpr int NO = 1;
    const int vx = tx, vy = ty;
    {
      float st[5][NO];
#pragma unroll
      for (int q = 0; q < 5; q++) {
        float acc[NO];
#pragma unroll
        for (int o = 0; o < NO; o++) acc[o] = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          acc[0] = fmaf(cfg.window[k], hS[q][vy + k][vx], acc[0]);
        }
#pragma unroll
        for (int o = 0; o < NO; o++) st[q][o] = acc[o];
      }
#pragma unroll
      for (int o = 0; o < NO; o++) {
        const int opx = x0 + vx + o, opy = y0 + vy;
        if (opx < cfg.W && opy < cfg.H) {
          const uint32_t opix = (uint32_t)opy * (uint32_t)cfg.W + (uint32_t)opx;
          const float m1 = st[0][o], m2 = st[1][o];
          const float s1 = st[2][o] - m1 * m1, s2 = st[3][o] - m2 * m2, s12 = st[4][o] - m1 * m2;
          const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s1 + s2 + C2;
          // v_rcp_f32 (1 ulp) instead of three IEEE divisions (their scale / fixup sequences held ~40 registers live)
          const float rB1 = __builtin_amdgcn_rcpf(B1), rB2 = __builtin_amdgcn_rcpf(B2);
          const float inv = rB1 * rB2;
          const float f = A1 * A2 * inv;
          ssim_sum += f;
          const float df_dm1 = 2.f * m2 * A2 * inv - f * 2.f * m1 * rB1;
          const float df_ds1 = -f * rB2;
          const float df_ds12 = 2.f * A1 * inv;
          bst(r_dm, (uint32_t)(ch * 3 + 0) * HW + opix, df_dm1 - 2.f * m1 * df_ds1 - m2 * df_ds12);  // d/d mu1 (total)
          bst(r_dm, (uint32_t)(ch * 3 + 1) * HW + opix, df_ds1);                                     // d/d E[x^2]
          bst(r_dm, (uint32_t)(ch * 3 + 2) * HW + opix, df_ds12);                                    // d/d E[xy]
        }
      }
    }
  }
  if constexpr (ROWS) {
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = 0.0;
    if (inside) {
      // (the L1 sum was taken from the staged channels; loss_px_sums only needs it and the depth / silhouette / reference)
      const float sil = out[4 * HW + pix];
      const float rgb[3] = {l1, 0.f, 0.f}, g3[3] = {0.f, 0.f, 0.f};
      const bool var = loss_variant(cfg), need = cfg.w_pearson != 0.f || var;
      const float depth = need ? out[3 * HW + pix] : 0.f, refv = (need && ref) ? ref[pix] : 0.f;
      loss_px_sums(cfg, rgb, sil, depth, g3, refv, acc);
      if (var) variant_px_sums(cfg, l1, sil, depth, refv, acc);
    }
    acc[2] = (double)ssim_sum;
    block_sums<12>(acc, red, loss_double_cols(cfg));
    if (tid == 0) {
      double* row = partial + (size_t)tile * 12;
#pragma unroll
      for (int k = 0; k < 12; k++) row[k] = acc[k];
    }
  } else {
    // only the tile's SSIM sum (column 2 of its row; the other columns came from the forward compositor)
    const float t = wave_sum_to_lane63(ssim_sum);
    if ((tid & 63) == 63) red[tid >> 6][0] = (double)t;
    __syncthreads();
    if (tid == 0) partial[(size_t)tile * 12 + 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  }
}

// per-pixel sums only (no SSIM term): one row of partial sums per 16x16 tile (plain stores: 14k double atomics on two cache
// lines cost ~90 us)
__global__ void __launch_bounds__(256)
loss_rows_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                 double* __restrict__ partial, int tiles_x) {
  __shared__ double red[4][12];
  const int tile = blockIdx.x;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int px = (tile % tiles_x) * LT + tx, py = (tile / tiles_x) * LT + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const size_t HW = (size_t)cfg.H * cfg.W, pix = (size_t)py * cfg.W + px;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  if (inside) {
    const float rgb[3] = {out[pix], out[HW + pix], out[2 * HW + pix]}, g3[3] = {gt[pix], gt[HW + pix], gt[2 * HW + pix]};
    const bool var = loss_variant(cfg), need = cfg.w_pearson != 0.f || var;
    const float sil = out[4 * HW + pix], depth = need ? out[3 * HW + pix] : 0.f, refv = (need && ref) ? ref[pix] : 0.f;
    loss_px_sums(cfg, rgb, sil, depth, g3, refv, acc);
    if (var) variant_px_sums(cfg, fabsf(rgb[0] - g3[0]) + fabsf(rgb[1] - g3[1]) + fabsf(rgb[2] - g3[2]), sil, depth, refv, acc);
  }
  block_sums<12>(acc, red, loss_double_cols(cfg));
  if (threadIdx.x == 0) {
    double* row = partial + (size_t)tile * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) row[k] = acc[k];
  }
}

// one 1024-lane workgroup: lane = 16 * rowgroup + column; 64 row groups keep the dependent-load chains short (the
// kernel is pure latency), then the groups are added in a fixed order (deterministic).  Lane 0 also derives the Pearson
// scalars every pixel of loss_grad_kernel needs (sums[16..23]) and, when asked, the four loss values.
__global__ void __launch_bounds__(1024) loss_finish_kernel(LossCfg cfg, const double* __restrict__ partial, int nrows, double* __restrict__ sums,
                                                           float* __restrict__ loss4) {
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
    pearson_scalars(tot, cfg.w_pearson != 0.f ? 1 : 0, cfg.pearson_invert, sums);
    if (loss4) {
      if (loss_variant(cfg)) variant_scalars(cfg, sums, (size_t)cfg.H * cfg.W, loss4);
      else loss_scalars(cfg, sums, (size_t)cfg.H * cfg.W, loss4);
    }
  }
}

// ---- gradient image ------------------------------------------------------------------------------------------------------------
// write6: also store zeros to the silhouette / depth^2 planes (no loss term reaches them).  (In the mapping loop this whole pass
// runs inside the backward compositor's prologue instead: loss_tile.h, composite.hip.)
template <bool VAR>
__global__ void __launch_bounds__(256)
loss_grad_kernel(LossCfg cfg, const float* __restrict__ out, const float* __restrict__ gt, const float* __restrict__ ref,
                 const float* __restrict__ dmaps, const double* __restrict__ sums, float* __restrict__ dL, int tiles_x, int write6) {
  __shared__ __align__(16) LossGradSmem sm;
  float g4[4];
  bool inside;
  loss_grad_tile<VAR>(cfg, out, gt, ref, dmaps, sums, blockIdx.x, tiles_x, sm, g4, inside);
  if (inside) {
    const int px = (blockIdx.x % tiles_x) * LT + (threadIdx.x & 15), py = (blockIdx.x / tiles_x) * LT + (threadIdx.x >> 4);
    const size_t HW = (size_t)cfg.H * cfg.W, pix = (size_t)py * cfg.W + px;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) dL[ch * HW + pix] = g4[ch];
    if (write6) { dL[4 * HW + pix] = 0.f; dL[5 * HW + pix] = 0.f; }
  }
}

static int loss_tiles_x(const LossCfg& cfg) { return (cfg.W + LT - 1) / LT; }
static int loss_tiles(const LossCfg& cfg) { return loss_tiles_x(cfg) * ((cfg.H + LT - 1) / LT); }

void launch_loss_finish(const LossCfg& cfg, double* sums, const double* partial, hipStream_t s, float* loss4) {
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, s, cfg, partial, loss_tiles(cfg), sums, loss4);
}

// standalone form (mm3dgs_loss): rows (+ SSIM maps) -> finish (+ loss values) -> gradient image with all six planes written
void launch_loss(const LossCfg& cfg, const float* out, const float* gt, const float* ref, float* dmaps, double* sums, double* partial,
                 float* dL, float* loss, hipStream_t s) {
  const int T = loss_tiles(cfg), tx = loss_tiles_x(cfg);
  if (cfg.w_ssim != 0.f)
    hipLaunchKernelGGL(ssim_maps_kernel<true>, dim3(T), dim3(256), 0, s, cfg, out, gt, ref, dmaps, partial, sums, tx, T);
  else
    hipLaunchKernelGGL(loss_rows_kernel, dim3(T), dim3(256), 0, s, cfg, out, gt, ref, partial, tx);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, s, cfg, partial, T, sums, loss);
  const bool var = cfg.w_depth != 0.f || cfg.l1_sum != 0 || (cfg.l1_mask & 2) != 0;     // the `method: splatam` forms
  hipLaunchKernelGGL(var ? loss_grad_kernel<true> : loss_grad_kernel<false>, dim3(T), dim3(256), 0, s, cfg, out, gt, ref, dmaps, sums, dL, tx, 1);
}

// mapping-loop form (mm3dgs_slam_map): the forward compositor's epilogue already wrote the L1 / Pearson rows of this render.
// Two launches, no finishing launch; dL gets its first four planes only (the backward compositor is told not to read the rest).
// The loss values are produced on demand by launch_loss_finish (every row column is complete after the first launch here).
void launch_loss_after_forward_rows(const LossCfg& cfg, const float* out, const float* gt, const float* ref, float* dmaps, double* sums,
                                    double* partial, float* dL, hipStream_t s) {
  const int T = loss_tiles(cfg), tx = loss_tiles_x(cfg);
  hipLaunchKernelGGL(ssim_maps_kernel<false>, dim3(T + 1), dim3(256), 0, s, cfg, out, gt, ref, dmaps, partial, sums, tx, T);
  if (dL)   // NULL: the caller runs the gradient pass inside the backward compositor (composite.hip)
    hipLaunchKernelGGL(loss_grad_kernel<false>, dim3(T), dim3(256), 0, s, cfg, out, gt, ref, dmaps, sums, dL, tx, 0);
}
