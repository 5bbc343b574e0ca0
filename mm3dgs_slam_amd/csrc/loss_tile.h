// Tile-level pieces of the image-loss kernels shared between loss.hip and the backward compositor (composite.hip), which folds the
// gradient-image pass of the mapping loss into its prologue.
#pragma once
#include "mm3dgs_common.h"
#include "fused_api.h"
#include "loss_pixel.h"

#define LT 16
#define HALO 5
#define LW (LT + 2 * HALO)  // 26
#define SW 48               // LDS row stride of the 26-wide staging rows: the four rows a 32-lane group of the horizontal pass reads
                            // (8 two-output strips each, ds_read_b64) start 48 banks apart = on the four quarters of the 64 banks
#define HW_ 16              // LDS row stride of the 16-wide horizontal-pass outputs (dense): the two rows a 32-lane group of the
                            // vertical pass reads land on the two halves of the 32 banks

// Global accesses of the two image kernels go through buffer resources (SGPR descriptor + 32-bit VGPR byte offset): with flat
// 64-bit addresses the address pairs of the halo prefetch and of the map stores pushed ssim_maps_kernel past the 96 registers
// that 5 waves per SIMD allow (it spilled ~40 registers).
typedef __amdgpu_buffer_rsrc_t Rsrc;
__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float bld(Rsrc r, uint32_t elem) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, elem * 4u, 0, 0)); }
__device__ __forceinline__ void bst(Rsrc r, uint32_t elem, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), r, elem * 4u, 0, 0); }

struct HaloIdx { int off[3], lds[3]; };   // a lane's three elements of the 26x26 halo region: global offset (-1: outside the image), LDS index (-1: none)

__device__ __forceinline__ HaloIdx halo_index(const LossCfg& cfg, int x0, int y0) {
  HaloIdx h;
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const int i = threadIdx.x + e * 256;
    const int ly = i / LW, lx = i - ly * LW;
    const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
    const bool in = i < LW * LW && gx >= 0 && gx < cfg.W && gy >= 0 && gy < cfg.H;
    h.off[e] = in ? gy * cfg.W + gx : -1;
    h.lds[e] = i < LW * LW ? ly * SW + lx : -1;
  }
  return h;
}

// dL/d(rendered image) of the lane's pixel (tx = tid % 16, ty = tid / 16 of tile `tile`) for the four planes that receive a loss:
// adjoint 11x11 convolution of the SSIM derivative maps (one colour channel at a time through `sD` / `hD`: two-output strips
// horizontally, one output per lane vertically), L1 sign, Pearson gradient.  Every lane of the 256-lane workgroup must call it
// (barriers inside).  Used by loss_grad_kernel and, folded into its prologue, by the mapping-mode backward compositor.
// <VAR>: the `method: splatam` forms of the per-pixel terms (LossCfg: colour L1 over { ref > 0 } too, sums instead of means, depth-L1
// term); only loss_grad_kernel instantiates it -- the compositor's prologue is never given such a configuration (api.hip).
struct LossGradSmem { float sD[3][LW][SW]; float hD[3][LW][HW_]; };
__device__ __forceinline__ bool loss_mask_on(int bits, bool smask, float refv) { return (!(bits & 1) || smask) && (!(bits & 2) || refv > 0.f); }
template <bool VAR = false>
__device__ __forceinline__ void loss_grad_tile(const LossCfg& cfg, const float* __restrict__ out, const float* __restrict__ gt,
                                               const float* __restrict__ ref, const float* __restrict__ dmaps, const double* __restrict__ sums,
                                               int tile, int tiles_x, LossGradSmem& sm, float (&g4)[4], bool& inside_out) {
  float (*sD)[LW][SW] = sm.sD;
  float (*hD)[LW][HW_] = sm.hD;
  const int x0 = (tile % tiles_x) * LT, y0 = (tile / tiles_x) * LT;
  const int tid = threadIdx.x;
  const uint32_t HW = (uint32_t)cfg.H * (uint32_t)cfg.W;
  const float l1_scale = loss_l1_scale(cfg, sums);
  const float ssim_scale = -cfg.w_ssim / (float)(3.0 * (double)HW);
  const int tx = tid & 15, ty = tid >> 4;      // this lane's output pixel
  const int px = x0 + tx, py = y0 + ty;
  const bool inside = px < cfg.W && py < cfg.H;
  const uint32_t pix = (uint32_t)py * (uint32_t)cfg.W + (uint32_t)px;
  const Rsrc r_out = make_rsrc(out, HW * 24u), r_gt = make_rsrc(gt, HW * 12u);
  float oc[3] = {0.f, 0.f, 0.f}, gc[3] = {0.f, 0.f, 0.f}, sil = 0.f;
  if (inside) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) { oc[ch] = bld(r_out, (uint32_t)ch * HW + pix); gc[ch] = bld(r_gt, (uint32_t)ch * HW + pix); }
    sil = bld(r_out, 4u * HW + pix);
  }
  float gch[3] = {0.f, 0.f, 0.f};
  if (cfg.w_ssim != 0.f) {
    const HaloIdx hx = halo_index(cfg, x0, y0);
    const Rsrc r_dm = make_rsrc(dmaps, HW * 36u);
    float v[3][3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int e = 0; e < 3; e++) v[q][e] = hx.off[e] >= 0 ? bld(r_dm, (uint32_t)q * HW + (uint32_t)hx.off[e]) : 0.f;
#pragma unroll 1
    for (int ch = 0; ch < 3; ch++) {
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < 3; e++)
          if (hx.lds[e] >= 0) (&sD[q][0][0])[hx.lds[e]] = v[q][e];
      __syncthreads();   // also orders the previous channel's vertical pass before this channel's hD writes
      if (ch < 2) {
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
          for (int e = 0; e < 3; e++) v[q][e] = hx.off[e] >= 0 ? bld(r_dm, (uint32_t)((ch + 1) * 3 + q) * HW + (uint32_t)hx.off[e]) : 0.f;
      }
      // horizontal pass: 3 maps x 26 rows x 8 strips of 2 outputs = 624 items over the 256 lanes
      for (int it = tid; it < 3 * LW * 8; it += 256) {
        const int q = it / (LW * 8), rem = it - q * (LW * 8);
        const int row = rem >> 3, s = rem & 7;
        const float2* ra = (const float2*)&sD[q][row][2 * s];
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const float2 t = ra[u];
          if (2 * u < 11) r0 = fmaf(cfg.window[2 * u], t.x, r0);
          if (2 * u - 1 >= 0) r1 = fmaf(cfg.window[2 * u - 1], t.x, r1);
          if (2 * u + 1 < 11) r0 = fmaf(cfg.window[2 * u + 1], t.y, r0);
          r1 = fmaf(cfg.window[2 * u], t.y, r1);
        }
        *(float2*)&hD[q][row][2 * s] = make_float2(r0, r1);
      }
      __syncthreads();
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = cfg.window[k];
        c0 = fmaf(w, hD[0][ty + k][tx], c0); c1 = fmaf(w, hD[1][ty + k][tx], c1); c2 = fmaf(w, hD[2][ty + k][tx], c2);
      }
      const float o_ = ch == 0 ? oc[0] : (ch == 1 ? oc[1] : oc[2]), g_ = ch == 0 ? gc[0] : (ch == 1 ? gc[1] : gc[2]);
      const float gval = ssim_scale * (c0 + 2.f * o_ * c1 + g_ * c2);
      if (ch == 0) gch[0] = gval; else if (ch == 1) gch[1] = gval; else gch[2] = gval;
    }
  }
  g4[0] = g4[1] = g4[2] = g4[3] = 0.f;
  inside_out = inside;
  if (inside) {
    const bool smask = sil > cfg.sil_thr;
    if constexpr (VAR) {
      const float refv = ref ? ref[pix] : 0.f, depth = out[3 * HW + pix];
      const bool on = loss_mask_on(cfg.l1_mask, smask, refv);
      const float sc = cfg.l1_sum ? cfg.w_l1 : l1_scale;          // d/dx of w sum|x - g|  or of  w mean|x - g| (3 n elements)
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        const float d = oc[ch] - gc[ch];
        g4[ch] = gch[ch] + (on ? sc * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f);
      }
      g4[3] = cfg.w_pearson != 0.f ? loss_px_pearson_grad(cfg, sil, depth, refv, sums) : 0.f;
      if (cfg.w_depth != 0.f && loss_mask_on(cfg.depth_mask, smask, refv)) {
        // sums[3] = pixels of the depth mask, sums[4] = sum |ref - depth| (the Pearson columns: the two terms are exclusive)
        const float dsc = cfg.l1_sum ? cfg.w_depth : (sums[3] > 0.0 ? cfg.w_depth / (float)sums[3] : 0.f);
        const float d = depth - refv;
        g4[3] += dsc * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) g4[ch] = gch[ch] + loss_px_l1_grad(cfg, oc[ch], gc[ch], smask, l1_scale);
      // depth channel: Pearson; silhouette and depth^2 carry no loss
      g4[3] = cfg.w_pearson != 0.f ? loss_px_pearson_grad(cfg, sil, out[3 * HW + pix], ref[pix], sums) : 0.f;
    }
  }
}
