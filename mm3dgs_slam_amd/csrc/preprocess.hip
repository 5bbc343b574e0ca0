// Per-Gaussian stages of the rasterizer for gfx950: forward projection / EWA splat / SH colour / tile counting,
// and the backward chain rule from screen-space gradients to the Gaussian parameters and the camera.
// One lane per Gaussian, 256-lane workgroups (4 waves); inputs are the caller's torch layouts ([P,3], [P,4],
// [P,M,3] ...), a wave reads each of them as one contiguous span so every fetched line is fully used.
//
// Maths: SURVEY.md Appendix A.  Conventions pinned by the reference: quaternion (w,x,y,z) and R(q)
// (utils/general_utils.py:78-99), cov3D = (RS)(RS)^T (utils/general_utils.py:101-110), SH basis
// (utils/sh_utils.py:57-112, +0.5 and clamp at slam/renderer.py:188-189), row-vector matrices
// (slam/renderer.py:117-124).
#include "mm3dgs_common.h"

#include "mm3dgs_math.h"
#include "composite_common.h"

// SH rows as float4s: a lane's [M,3] coefficient row is 12 M contiguous bytes (192 B at degree 3), so with scalar accesses
// every one of its 3 M load (store) instructions touches 64 different cache lines for 4 useful bytes each.  With M in {4, 16}
// and a 16-byte aligned base the row is read / written as 3 M / 4 float4s per lane: a quarter of the line visits.
// (MV = M for this path, 0 = the scalar path for every other M or an unaligned base.)
template <int MV>
__device__ __forceinline__ void sh_row_load(const float* __restrict__ row, float (&v)[MV * 3]) {
  const float4* r = (const float4*)row;
#pragma unroll
  for (int j = 0; j < MV * 3 / 4; j++) {
    const float4 q = r[j];
    v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
  }
}
template <int MV>
__device__ __forceinline__ void sh_row_store(float* __restrict__ row, const float (&v)[MV * 3]) {
  float4* r = (float4*)row;
#pragma unroll
  for (int j = 0; j < MV * 3 / 4; j++) r[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

template <int MV>
__global__ void __launch_bounds__(PP_BLOCK)
preprocess_fwd_kernel(CamDev cam, int P, int M, int C, const float* __restrict__ means3D,
                      const float* __restrict__ shs, const float* __restrict__ colors,
                      const float* __restrict__ opac, const float* __restrict__ scales,
                      const float* __restrict__ rots, const float* __restrict__ cov3d, int32_t* __restrict__ radii,
                      GeomView g, ImageView iv, int lds_tiles) {
  // Per-workgroup tile histogram in LDS (lds_tiles = T when it fits, else 0 -> direct global atomics): consecutive
  // Gaussians are spatially coherent in SLAM maps (seeded in pixel raster order, slam/mapper.py:437-474), so a
  // workgroup's ~2.3 overlaps per Gaussian collapse into a handful of global atomics.
  extern __shared__ uint32_t hist[];
  const int T = cam.gx * cam.gy;
  for (int t = threadIdx.x; t < lds_tiles; t += PP_BLOCK) hist[t] = 0;
  if (lds_tiles) __syncthreads();
  int idx = blockIdx.x * PP_BLOCK + threadIdx.x;
  const bool live = idx < P;
  const float* V = cam.view;
  const float* PV = cam.proj;
  float p[3] = {0.f, 0.f, 0.f};
  if (live) { p[0] = means3D[(size_t)idx * 3]; p[1] = means3D[(size_t)idx * 3 + 1]; p[2] = means3D[(size_t)idx * 3 + 2]; }
  float tz = p[0] * V[2] + p[1] * V[6] + p[2] * V[10] + V[14];
  int32_t rad = 0;
  uint32_t r0 = 0, r1 = 0;
  if (live && tz > 0.2f) {
    float hx = p[0] * PV[0] + p[1] * PV[4] + p[2] * PV[8] + PV[12];
    float hy = p[0] * PV[1] + p[1] * PV[5] + p[2] * PV[9] + PV[13];
    float hw = p[0] * PV[3] + p[1] * PV[7] + p[2] * PV[11] + PV[15];
    float pw = 1.f / (hw + 1e-7f);
    float S3[3][3], R[3][3], sm[3];
    load_cov3d(idx, scales, rots, cov3d, cam.scale_modifier, S3, R, sm);
    Ewa e;
    ewa_project(cam, V, p, S3, e);
    float det = e.a * e.c - e.b * e.b;
    float px = ((hx * pw + 1.f) * cam.W - 1.f) * 0.5f;
    float py = ((hy * pw + 1.f) * cam.H - 1.f) * 0.5f;
    if (det != 0.f && isfinite(px) && isfinite(py)) {
      float dinv = 1.f / det;
      float mid = 0.5f * (e.a + e.c);
      float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      float rf = ceilf(3.f * sqrtf(lam));
      float gxf = (float)cam.gx + 1.f, gyf = (float)cam.gy + 1.f;
      int minx = min(cam.gx, max(0, (int)fminf(fmaxf((px - rf) / TILE, -1.f), gxf)));
      int miny = min(cam.gy, max(0, (int)fminf(fmaxf((py - rf) / TILE, -1.f), gyf)));
      int maxx = min(cam.gx, max(0, (int)fminf(fmaxf((px + rf + (TILE - 1)) / TILE, -1.f), gxf)));
      int maxy = min(cam.gy, max(0, (int)fminf(fmaxf((py + rf + (TILE - 1)) / TILE, -1.f), gyf)));
      if ((maxx - minx) * (maxy - miny) > 0) {
        rad = (int32_t)fminf(rf, 2.0e9f);
        r0 = (uint32_t)minx | ((uint32_t)miny << 16);
        r1 = (uint32_t)maxx | ((uint32_t)maxy << 16);
        float col[MM3DGS_MAX_CHANNELS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int nsh = 0;
        if (shs) {
          nsh = 3;
          float dx = p[0] - cam.campos[0], dy = p[1] - cam.campos[1], dz = p[2] - cam.campos[2];
          float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
          float b[16];
          sh_basis(cam.sh_degree, dx * inv, dy * inv, dz * inv, b);
          int nb = (cam.sh_degree + 1) * (cam.sh_degree + 1);
          const float* sh = shs + (size_t)idx * M * 3;
          float c0 = 0.f, c1 = 0.f, c2 = 0.f;
          if (MV) {
            float sv[(MV ? MV : 4) * 3];
            sh_row_load<(MV ? MV : 4)>(sh, sv);
#pragma unroll
            for (int k = 0; k < MV; k++)
              if (k < nb) { c0 += b[k] * sv[k * 3]; c1 += b[k] * sv[k * 3 + 1]; c2 += b[k] * sv[k * 3 + 2]; }
          } else {
            for (int k = 0; k < nb; k++) { c0 += b[k] * sh[k * 3]; c1 += b[k] * sh[k * 3 + 1]; c2 += b[k] * sh[k * 3 + 2]; }
          }
          c0 += 0.5f; c1 += 0.5f; c2 += 0.5f;
          uint8_t cl = (c0 < 0.f ? 1 : 0) | (c1 < 0.f ? 2 : 0) | (c2 < 0.f ? 4 : 0);
          g.clamped[idx] = cl;
          col[0] = fmaxf(c0, 0.f); col[1] = fmaxf(c1, 0.f); col[2] = fmaxf(c2, 0.f);
        }
        if (colors) {
          int ne = C - nsh;
          for (int k = 0; k < ne; k++) col[nsh + k] = colors[(size_t)idx * ne + k];
        }
        float4* sp = (float4*)(g.splat + (size_t)idx * SPLAT_F);
        const float4 sA = make_float4(px, py, e.c * dinv, -e.b * dinv), sB = make_float4(e.a * dinv, opac[idx], col[0], col[1]);
        sp[0] = sA;
        sp[1] = sB;
        sp[2] = make_float4(col[2], col[3], col[4], col[5]);
        g.depth[idx] = e.t[2];
      }
    }
  }
  if (live) {
    radii[idx] = rad;
    g.rect[(size_t)idx * 2] = r0;
    g.rect[(size_t)idx * 2 + 1] = r1;
  }
  // count overlaps per tile
  {
    const int minx = r0 & 0xffff, miny = r0 >> 16, maxx = r1 & 0xffff, maxy = r1 >> 16;
    const int w = maxx - minx, area = w * (maxy - miny);
    {  // workgroup-local exclusive scan of tiles touched (this Gaussian's first pair index)
      __shared__ uint32_t wtot[PP_BLOCK / 64];
      const int ln = threadIdx.x & 63, wvi = threadIdx.x >> 6;
      // (the second scan -- the 4x4 blocks of every splat's block rectangle, blkoff / block_blk: the first Gaussian-major block record -- left in round 6:
      //  block records are addressed by list position in every mode)
      const uint32_t x = wave_scan_incl((uint32_t)area);   // tiles touched
      if (ln == 63) wtot[wvi] = x;
      __syncthreads();
      uint32_t pre = 0;
      for (int q = 0; q < wvi; q++) pre += wtot[q];
      if (live) g.tileoff[idx] = pre + x - (uint32_t)area;
      if (threadIdx.x == PP_BLOCK - 1) g.block_tiles[blockIdx.x] = pre + x;
    }
    uint32_t* cnt = lds_tiles ? hist : iv.tile_count;
    const int lane = threadIdx.x & 63;
    unsigned long long big = __ballot(area > 32);
    if (area > 0 && area <= 32)
      for (int y = miny; y < maxy; y++)
        for (int x = minx; x < maxx; x++) atomicAdd(&cnt[y * cam.gx + x], 1u);
    while (big) {  // a splat covering many tiles is spread over the whole wave
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const int sminx = __builtin_amdgcn_readlane(minx, src), sminy = __builtin_amdgcn_readlane(miny, src);
      const int sw = __builtin_amdgcn_readlane(w, src), sarea = __builtin_amdgcn_readlane(area, src);
      for (int k = lane; k < sarea; k += 64) atomicAdd(&cnt[(sminy + k / sw) * cam.gx + sminx + k % sw], 1u);
    }
  }
  if (lds_tiles) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += PP_BLOCK) {
      uint32_t c = hist[t];
      if (c) atomicAdd(&iv.tile_count[t], c);
    }
  }
}

void launch_preprocess_fwd(const CamDev& cam, int P, int M, int C, const float* means3D, const float* shs,
                           const float* colors, const float* opac, const float* scales, const float* rots,
                           const float* cov3d, int32_t* radii, GeomView g, ImageView iv, hipStream_t s) {
  if (P <= 0) return;
  const int T = cam.gx * cam.gy;
  const int lds_tiles = T <= MAX_LDS_TILES ? T : 0;
  const bool vec = shs && (((uintptr_t)shs & 15) == 0);
  auto kern = (vec && M == 16) ? preprocess_fwd_kernel<16> : ((vec && M == 4) ? preprocess_fwd_kernel<4> : preprocess_fwd_kernel<0>);
  hipLaunchKernelGGL(kern, dim3((P + PP_BLOCK - 1) / PP_BLOCK), dim3(PP_BLOCK), (size_t)lds_tiles * 4, s, cam,
                     P, M, C, means3D, shs, colors, opac, scales, rots, cov3d, radii, g, iv, lds_tiles);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward: screen-space gradient record (dxy in pixel units, dconic, dopacity, dcolour[6]) -> parameter gradients.
// Camera gradients: 27 values per Gaussian (view rows 0..3 x cols 0..2, proj rows 0..3 x cols {0,1,3}, campos)
// are reduced wave -> workgroup in registers/LDS, one partial row per workgroup is written, and a single-wave
// finishing kernel adds the rows in double precision in a fixed order (deterministic, no atomics).
// Sum of a Gaussian's per-tile gradient records (generic mode: 6 + C floats at a packed stride of `recf` floats).  Every lane of the wave must call
// it.  Rectangles of up to 16 tiles are summed by their own lane, four records in flight, in ascending pair order; a bigger one is read by the whole
// wave (lane-strided, then a fixed-order reduction) -- deterministic either way.  (fused.hip's gather_tile_records is the SLAM modes' instance.)
__device__ __forceinline__ void gather_pair_records(int area, uint32_t first, const float* __restrict__ dtile, int recf, float4& acc0, float4& acc1,
                                                    float4& acc2) {
  const int lane = threadIdx.x & 63;
  const bool big = area > 16;
  const int n_own = big ? 0 : area;
  for (int k0 = 0; __ballot(k0 < n_own) != 0ull; k0 += 4) {
    float4 a[4], b4[4], c4[4];
    bool on[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      on[u] = k0 + u < n_own;
      const float* r = dtile + (on[u] ? (size_t)(first + (uint32_t)(k0 + u)) * recf : (size_t)0);
      a[u] = ld4u(r); b4[u] = ld4u(r + 4); c4[u] = ld4u(r + 8);      // (a shorter record: the caller drops what belongs to the next one)
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      acc0.x += on[u] ? a[u].x : 0.f; acc0.y += on[u] ? a[u].y : 0.f; acc0.z += on[u] ? a[u].z : 0.f; acc0.w += on[u] ? a[u].w : 0.f;
      acc1.x += on[u] ? b4[u].x : 0.f; acc1.y += on[u] ? b4[u].y : 0.f; acc1.z += on[u] ? b4[u].z : 0.f; acc1.w += on[u] ? b4[u].w : 0.f;
      acc2.x += on[u] ? c4[u].x : 0.f; acc2.y += on[u] ? c4[u].y : 0.f; acc2.z += on[u] ? c4[u].z : 0.f; acc2.w += on[u] ? c4[u].w : 0.f;
    }
  }
  for (unsigned long long bigs = __ballot(big); bigs; bigs &= bigs - 1ull) {
    const int src = __ffsll((long long)bigs) - 1;
    const int sarea = __builtin_amdgcn_readlane(area, src);
    const uint32_t sfirst = (uint32_t)__builtin_amdgcn_readlane((int)first, src);
    float v[12];
#pragma unroll
    for (int f = 0; f < 12; f++) v[f] = 0.f;
    for (int k = lane; k < sarea; k += 64) {
      const float* r = dtile + (size_t)(sfirst + (uint32_t)k) * recf;
#pragma unroll
      for (int f = 0; f < 12; f++) v[f] += f < recf ? r[f] : 0.f;
    }
#pragma unroll
    for (int f = 0; f < 12; f++) v[f] = wave_sum(v[f]);
    if (lane == src) {
      acc0 = make_float4(v[0], v[1], v[2], v[3]); acc1 = make_float4(v[4], v[5], v[6], v[7]); acc2 = make_float4(v[8], v[9], v[10], v[11]);
    }
  }
}

#define NCAM 27
// (developer timing probes, variant builds only -- results INVALID: -DMM3DGS_PPB_PROBE=<bits>  1: no record gather | 2: no dL/dSH store | 4: no SH row load)
#ifndef MM3DGS_PPB_PROBE
#define MM3DGS_PPB_PROBE 0
#endif
template <int MV>
__global__ void __launch_bounds__(PP_BLOCK)
preprocess_bwd_kernel(CamDev cam, int P, int M, int C, const float* __restrict__ means3D,
                      const float* __restrict__ shs, const float* __restrict__ colors,
                      const float* __restrict__ opac, const float* __restrict__ scales,
                      const float* __restrict__ rots, const float* __restrict__ cov3d,
                      const int32_t* __restrict__ radii, GeomView g, BinView bn, uint32_t N_cap,
                      const float* __restrict__ dsub, float* __restrict__ campartial, float* __restrict__ dmeans3D, float* __restrict__ dmeans2D,
                      float* __restrict__ dshs, float* __restrict__ dcolors, float* __restrict__ dopac,
                      float* __restrict__ dscales, float* __restrict__ drots, float* __restrict__ dcov3d,
                      int want_cam, int flags) {
  int idx = blockIdx.x * PP_BLOCK + threadIdx.x;
  const float* V = cam.view;
  const float* PV = cam.proj;
  float cg[NCAM];
#pragma unroll
  for (int k = 0; k < NCAM; k++) cg[k] = 0.f;
  const bool skip_g = (flags & MM3DGS_BWD_SKIP_GAUSSIAN_GRADS) != 0;
  const int nsh = shs ? 3 : 0;
  const int ne = C - nsh;
  // ---- gather this Gaussian's screen-space gradient: sum of the records its (block, splat) pairs wrote, in a
  // fixed order (deterministic).  A Gaussian covering more than 32 tiles is summed by the whole wave.
  float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0, acc2 = acc0;
  {
    // one round of independent loads, then the sum of this Gaussian's per-tile records (composite.hip's per-tile combine wrote one record per
    // (tile, splat) pair at the pair's Gaussian-major index: a Gaussian's pairs -- its tile rectangle, row-major -- are one contiguous span, and so
    // are the spans of consecutive Gaussians).  Round 6: before, this kernel walked the (4x4 block, splat) records of every pair through their block
    // masks -- four times the records, 640 of its 990 us at 1080p / 3 M Gaussians (profiles/r06_c5_probes.txt).
    uint32_t first = 0;
    int area = 0;
    if (idx < P) {
      const uint32_t r0 = g.rect[(size_t)idx * 2], r1 = g.rect[(size_t)idx * 2 + 1];
      const uint32_t toff = g.tileoff[idx], btile = g.block_tiles[idx >> 8];
      if (r1 != r0) {   // <=> radii > 0
        area = ((int)(r1 & 0xffff) - (int)(r0 & 0xffff)) * ((int)(r1 >> 16) - (int)(r0 >> 16));
        first = btile + toff;
        if ((size_t)first + (size_t)area > (size_t)N_cap) area = 0;      // beyond the capacity (flagged by the forward): nothing was written
      }
    }
    if (!(MM3DGS_PPB_PROBE & 1))
    gather_pair_records(area, first, dsub + (size_t)NLIST * (size_t)N_cap * SPLAT_F, GENERIC_RECF(C), acc0, acc1, acc2);
    // a record shorter than 12 floats: what the last float4 picked up past its end belongs to the next record
    if (C < 6) { if (C < 3) { acc2.x = 0.f; } if (C < 4) acc2.y = 0.f; if (C < 5) acc2.z = 0.f; acc2.w = 0.f; if (C < 2) acc1.w = 0.f; if (C < 1) acc1.z = 0.f; }
  }
  float shg0 = 0.f, shg1 = 0.f, shg2 = 0.f, shu0 = 0.f, shu1 = 0.f, shu2 = 1.f;      // (MV == 16) clamp-masked colour gradient and viewing direction of this lane's Gaussian: zero gradient when culled
  if (idx < P) {
    float dmean[3] = {0.f, 0.f, 0.f};
    float gnx = 0.f, gny = 0.f;
    float ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f}, dc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dop = 0.f;
    float dcol[MM3DGS_MAX_CHANNELS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool vis = radii[idx] > 0;
    if (vis) {
      float4 d0 = acc0, d1 = acc1, d2 = acc2;
      // moments -> d/dxy (pixel units) and d/dconic, with this splat's conic (composite.hip record layout)
      const float4* spl = (const float4*)(g.splat + (size_t)idx * SPLAT_F);
      const float4 sp0 = spl[0], sp1 = spl[1];
      const float qa = sp0.z, qb = sp0.w, qc = sp1.x;
      float gpx = -(qa * d0.x + qb * d0.y), gpy = -(qc * d0.y + qb * d0.x);
      float gA = -0.5f * d0.z, gB = -d0.w, gC = -0.5f * d1.x;
      dop = d1.y;
      dcol[0] = d1.z; dcol[1] = d1.w; dcol[2] = d2.x; dcol[3] = d2.y; dcol[4] = d2.z; dcol[5] = d2.w;
      float p[3] = {means3D[(size_t)idx * 3], means3D[(size_t)idx * 3 + 1], means3D[(size_t)idx * 3 + 2]};
      float S3[3][3], R[3][3], sm[3];
      load_cov3d(idx, scales, rots, cov3d, cam.scale_modifier, S3, R, sm);
      Ewa e;
      ewa_project(cam, V, p, S3, e);
      float a = e.a, b = e.b, c = e.c;
      float det = a * c - b * b;
      float idet = 1.f / det;
      // conic = (c, -b, a)/det  ->  d/d(a,b,c)
      // (G2 as 1/det [[gC, -gB/2], [-gB/2, gA]] + kappa adj(Sigma2), not the expanded closed form: fused.hip slam_bwd_body says why)
      float kappa = -(c * gA - b * gB + a * gC) * idet * idet;
      float da = gC * idet + kappa * c;
      float db = -gB * idet - 2.f * (kappa * b);
      float dcc = gA * idet + kappa * a;
      float G2[2][2] = {{da, 0.5f * db}, {0.5f * db, dcc}};
      // GA = G2 * A (2x3)
      float GA[2][3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        GA[0][i] = G2[0][0] * e.A[0][i] + G2[0][1] * e.A[1][i];
        GA[1][i] = G2[1][0] * e.A[0][i] + G2[1][1] * e.A[1][i];
      }
      // dSigma3 = A^T G2 A (symmetric)
      float dS[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) dS[i][j] = e.A[0][i] * GA[0][j] + e.A[1][i] * GA[1][j];
      // dSigma3 is symmetric in exact arithmetic; in float32 the two triangles round differently, and their difference IS the rotation
      // gradient of an isotropic Gaussian with identity rotation (every freshly seeded one, slam/mapper.py:644-668): rounding noise that
      // Adam(eps=1e-15) turns into full +-lr steps of the quaternion.  torch's autograd of Sigma = L L^T forms (dSigma + dSigma^T) L and
      // is exactly zero there; so is this once the triangles are averaged (found with the G9 runs of the reference's own classes:
      // pose error to the reference 1.4e-5 -> 8e-8 after the first tracked frame).
      {
        const float s01 = 0.5f * (dS[0][1] + dS[1][0]), s02 = 0.5f * (dS[0][2] + dS[2][0]), s12 = 0.5f * (dS[1][2] + dS[2][1]);
        dS[0][1] = s01; dS[1][0] = s01; dS[0][2] = s02; dS[2][0] = s02; dS[1][2] = s12; dS[2][1] = s12;
      }
      // dA = 2 G2 A Sigma3
      float dA[2][3];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) dA[r][j] = 2.f * (GA[r][0] * S3[0][j] + GA[r][1] * S3[1][j] + GA[r][2] * S3[2][j]);
      // dJ = dA Wr^T : dJ[r][k] = sum_i dA[r][i] * Wr[k][i] = sum_i dA[r][i] * V[i][k]
      float dJ00 = dA[0][0] * V[0] + dA[0][1] * V[4] + dA[0][2] * V[8];
      float dJ02 = dA[0][0] * V[2] + dA[0][1] * V[6] + dA[0][2] * V[10];
      float dJ11 = dA[1][0] * V[1] + dA[1][1] * V[5] + dA[1][2] * V[9];
      float dJ12 = dA[1][0] * V[2] + dA[1][1] * V[6] + dA[1][2] * V[10];
      float tz = e.t[2], itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
      float dtxc = -cam.focal_x * itz2 * dJ02;
      float dtyc = -cam.focal_y * itz2 * dJ12;
      float dt[3];
      dt[0] = e.in_x ? dtxc : 0.f;
      dt[1] = e.in_y ? dtyc : 0.f;
      dt[2] = -cam.focal_x * itz2 * dJ00 - cam.focal_y * itz2 * dJ11 + 2.f * cam.focal_x * e.txc * itz3 * dJ02 +
              2.f * cam.focal_y * e.tyc * itz3 * dJ12;
#pragma unroll
      for (int i = 0; i < 3; i++) dmean[i] = V[i * 4 + 0] * dt[0] + V[i * 4 + 1] * dt[1] + V[i * 4 + 2] * dt[2];
      // screen position: pix = ((ndc+1) S - 1)/2, ndc = hom.xy / (hom.w + 1e-7)
      float hx = p[0] * PV[0] + p[1] * PV[4] + p[2] * PV[8] + PV[12];
      float hy = p[0] * PV[1] + p[1] * PV[5] + p[2] * PV[9] + PV[13];
      float hw = p[0] * PV[3] + p[1] * PV[7] + p[2] * PV[11] + PV[15];
      float pw = 1.f / (hw + 1e-7f);
      gnx = gpx * 0.5f * cam.W;
      gny = gpy * 0.5f * cam.H;
      float dhx = gnx * pw, dhy = gny * pw, dhw = -(gnx * hx + gny * hy) * pw * pw;
#pragma unroll
      for (int i = 0; i < 3; i++) dmean[i] += PV[i * 4 + 0] * dhx + PV[i * 4 + 1] * dhy + PV[i * 4 + 3] * dhw;
      if (want_cam) {
        // view: from t = [p,1] V  and from Wr inside A = J Wr  (dWr = J^T dA ; V[i][j] = Wr[j][i])
        float ph[4] = {p[0], p[1], p[2], 1.f};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) cg[i * 3 + j] = ph[i] * dt[j];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          cg[i * 3 + 0] += e.J00 * dA[0][i];
          cg[i * 3 + 1] += e.J11 * dA[1][i];
          cg[i * 3 + 2] += e.J02 * dA[0][i] + e.J12 * dA[1][i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          cg[12 + i * 3 + 0] = ph[i] * dhx;
          cg[12 + i * 3 + 1] = ph[i] * dhy;
          cg[12 + i * 3 + 2] = ph[i] * dhw;
        }
      }
      // colours
      if (shs) {
        uint8_t cl = g.clamped[idx];
        float gc0 = (cl & 1) ? 0.f : dcol[0], gc1 = (cl & 2) ? 0.f : dcol[1], gc2 = (cl & 4) ? 0.f : dcol[2];
        float vx = p[0] - cam.campos[0], vy = p[1] - cam.campos[1], vz = p[2] - cam.campos[2];
        float inv = 1.f / sqrtf(vx * vx + vy * vy + vz * vz);
        float ux = vx * inv, uy = vy * inv, uz = vz * inv;
        int deg = cam.sh_degree;
        int nb = (deg + 1) * (deg + 1);
        if (MV == 16) { shg0 = gc0; shg1 = gc1; shg2 = gc2; shu0 = ux; shu1 = uy; shu2 = uz; }      // stored by the whole wave below
        if (MV != 16 && !skip_g && dshs && !((MM3DGS_PPB_PROBE & 2) && gc0 != 123.f)) {
          float bb[16];
          sh_basis(deg, ux, uy, uz, bb);
          float* o = dshs + (size_t)idx * M * 3;
          if (MV) {
            float ov[(MV ? MV : 4) * 3];
#pragma unroll
            for (int k = 0; k < (MV ? MV : 4); k++) {
              const float bk = k < nb ? bb[k] : 0.f;
              ov[k * 3] = bk * gc0; ov[k * 3 + 1] = bk * gc1; ov[k * 3 + 2] = bk * gc2;
            }
            sh_row_store<(MV ? MV : 4)>(o, ov);
          } else {
            for (int k = 0; k < nb; k++) { o[k * 3] = bb[k] * gc0; o[k * 3 + 1] = bb[k] * gc1; o[k * 3 + 2] = bb[k] * gc2; }
            for (int k = nb; k < M; k++) { o[k * 3] = 0.f; o[k * 3 + 1] = 0.f; o[k * 3 + 2] = 0.f; }
          }
        }
        if (deg > 0 && !((MM3DGS_PPB_PROBE & 4) && gc0 != 123.f)) {
          float bx[16], by[16], bz[16];
          sh_basis_grad(deg, ux, uy, uz, bx, by, bz);
          const float* sh = shs + (size_t)idx * M * 3;
          // in double: the tangential projection below is a small difference of large terms, and its sum over the map is the
          // camera-position gradient (held to 1e-5; the float version sat at 1.2e-5 on the degree-2 case)
          double ddx = 0.0, ddy = 0.0, ddz = 0.0;
          if (MV) {
            float sv[(MV ? MV : 4) * 3];
            sh_row_load<(MV ? MV : 4)>(sh, sv);
#pragma unroll
            for (int k = 1; k < (MV ? MV : 4); k++)
              if (k < nb) {
                const double w = (double)sv[k * 3] * gc0 + (double)sv[k * 3 + 1] * gc1 + (double)sv[k * 3 + 2] * gc2;
                ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;
              }
          } else {
            for (int k = 1; k < nb; k++) {
              const double w = (double)sh[k * 3] * gc0 + (double)sh[k * 3 + 1] * gc1 + (double)sh[k * 3 + 2] * gc2;
              ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;
            }
          }
          const double dot = ux * ddx + uy * ddy + uz * ddz;
          float mx = (float)((ddx - ux * dot) * inv), my = (float)((ddy - uy * dot) * inv), mz = (float)((ddz - uz * dot) * inv);
          dmean[0] += mx; dmean[1] += my; dmean[2] += mz;
          if (want_cam) { cg[24] = -mx; cg[25] = -my; cg[26] = -mz; }
        }
      }
      // covariance parameters
      if (!skip_g) {
        if (cov3d) {
          dc6[0] = dS[0][0]; dc6[1] = 2.f * dS[0][1]; dc6[2] = 2.f * dS[0][2];
          dc6[3] = dS[1][1]; dc6[4] = 2.f * dS[1][2]; dc6[5] = dS[2][2];
        } else {
          // Sigma3 = Mx Mx^T, Mx = R diag(sm):  dMx = 2 dS Mx
          float dM[3][3];
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++)
              dM[i][k] = 2.f * (dS[i][0] * R[0][k] + dS[i][1] * R[1][k] + dS[i][2] * R[2][k]) * sm[k];
          float dR[3][3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            ds[k] = cam.scale_modifier * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
#pragma unroll
            for (int i = 0; i < 3; i++) dR[i][k] = dM[i][k] * sm[k];
          }
          float r = rots[(size_t)idx * 4], x = rots[(size_t)idx * 4 + 1], y = rots[(size_t)idx * 4 + 2], z = rots[(size_t)idx * 4 + 3];
          dq[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
          dq[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] +
                         r * dR[2][1] - 2.f * x * dR[2][2]);
          dq[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] +
                         z * dR[2][1] - 2.f * y * dR[2][2]);
          dq[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] +
                         y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
      }
    } else if (MV != 16 && !skip_g && shs && dshs) {
      float* o = dshs + (size_t)idx * M * 3;
      if (MV) {
#pragma unroll
        for (int j = 0; j < (MV ? MV : 4) * 3 / 4; j++) ((float4*)o)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        for (int k = 0; k < M * 3; k++) o[k] = 0.f;
      }
    }
    // every Gaussian writes its slots (culled ones write zeros): no memset of the outputs is needed
    if (dmeans3D) { dmeans3D[(size_t)idx * 3] = dmean[0]; dmeans3D[(size_t)idx * 3 + 1] = dmean[1]; dmeans3D[(size_t)idx * 3 + 2] = dmean[2]; }
    if (dmeans2D) { dmeans2D[(size_t)idx * 3] = gnx; dmeans2D[(size_t)idx * 3 + 1] = gny; dmeans2D[(size_t)idx * 3 + 2] = 0.f; }
    if (dcolors) {      // (static indices: a run-time index into dcol[] sends the array through scratch memory)
#pragma unroll
      for (int k = 0; k < MM3DGS_MAX_CHANNELS; k++)
        if (k < ne) dcolors[(size_t)idx * ne + k] = nsh ? dcol[k + 3 < MM3DGS_MAX_CHANNELS ? k + 3 : 0] : dcol[k];
    }
    if (!skip_g) {
      if (dopac) dopac[idx] = dop;
      if (dscales && !cov3d) { dscales[(size_t)idx * 3] = ds[0]; dscales[(size_t)idx * 3 + 1] = ds[1]; dscales[(size_t)idx * 3 + 2] = ds[2]; }
      if (drots && !cov3d) { drots[(size_t)idx * 4] = dq[0]; drots[(size_t)idx * 4 + 1] = dq[1]; drots[(size_t)idx * 4 + 2] = dq[2]; drots[(size_t)idx * 4 + 3] = dq[3]; }
      if (dcov3d && cov3d) for (int k = 0; k < 6; k++) dcov3d[(size_t)idx * 6 + k] = dc6[k];
    }
  }
  if constexpr (MV == 16) {
    // dL/dSH rows of the wave's 64 Gaussians, stored by the wave TOGETHER (round 6).  A lane's row is 192 contiguous bytes, and twelve float4 stores per
    // lane -- every one of them 64 lanes x 16 bytes in 64 different cache lines -- wrote the 576 MB of a 3 M-Gaussian map at half the rate a streaming
    // store reaches (140 of this kernel's 990 us at 1080p, profiles/r06_c5_probes.txt).  The rows go through a wave-private LDS tile in two halves of
    // 24 floats (7 KB per wave: the occupancy is the registers' either way): lane g writes its half row, then lane l stores the float4s l, l + 64, ...
    // of the tile -- 96-byte runs, six lanes to a row.  Culled Gaussians store zeros (no memset of the output is needed), rows beyond P are skipped.
    if (!skip_g && shs && dshs && !(MM3DGS_PPB_PROBE & 2)) {
      constexpr int HROW = 28;      // floats per half row in LDS: 24 + 4 (16-byte aligned rows, the stride spreads the banks)
      __shared__ __align__(16) float shx[PP_BLOCK / 64][64 * HROW];
      const int lane = threadIdx.x & 63, wvq = threadIdx.x >> 6;
      float bb[16];
      sh_basis(cam.sh_degree, shu0, shu1, shu2, bb);
      const int nb = (cam.sh_degree + 1) * (cam.sh_degree + 1);
      const size_t row0 = (size_t)blockIdx.x * PP_BLOCK + (size_t)wvq * 64;      // first Gaussian of this wave
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float4* mine = (float4*)(shx[wvq] + lane * HROW);
#pragma unroll
        for (int j = 0; j < 6; j++) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int e = 24 * h + 4 * j + c, k = e / 3, ch = e % 3;      // element e of the row = coefficient k, channel ch
            v[c] = (k < nb ? bb[k] : 0.f) * (ch == 0 ? shg0 : (ch == 1 ? shg1 : shg2));
          }
          mine[j] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 6; j++) {
          const int f = lane + 64 * j, r = f / 6, c4 = f % 6;      // float4 c4 of half row r
          const float4 q = *(const float4*)(shx[wvq] + r * HROW + c4 * 4);
          if (row0 + (size_t)r < (size_t)P) *(float4*)(dshs + (row0 + (size_t)r) * 48 + 24 * h + c4 * 4) = q;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (want_cam) {
    // double precision from the per-Gaussian terms up: the camera gradient is a sum of P terms of both signs, and the float
    // error of workgroup-level partial sums showed up as 2.6e-5 on dL/dproj (bar: 1e-5); not a hot path (the SLAM loops
    // take the pose gradient through slam_preprocess_bwd_kernel)
    __shared__ double red[4][NCAM];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NCAM; k++) {
      double v = wave_sum_to_lane63_f64((double)cg[k]);
      if (lane == 63) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NCAM) {
      int k = threadIdx.x;
      ((double*)campartial)[(size_t)blockIdx.x * 32 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
  }
}

// 256 lanes: lane = 32 * rowgroup + column (27 used); strided row sums, then the 8 row groups are added in a fixed order
// in double precision (deterministic).
__global__ void __launch_bounds__(256)
camgrad_finish_kernel(const double* __restrict__ campartial, int nrows, float* __restrict__ dview, float* __restrict__ dproj,
                      float* __restrict__ dcampos) {
  __shared__ double part[8][32];
  const int t = threadIdx.x, col = t & 31, grp = t >> 5;
  double acc = 0.0;
  if (col < NCAM)
    for (int r = grp; r < nrows; r += 8) acc += campartial[(size_t)r * 32 + col];
  part[grp][col] = acc;
  __syncthreads();
  const int k = t;
  if (k < NCAM) {
    double tot = 0.0;
    for (int q = 0; q < 8; q++) tot += part[q][k];
    const float v = (float)tot;
    if (k < 12) {
      int i = k / 3, j = k % 3;
      if (dview) dview[i * 4 + j] = v;
    } else if (k < 24) {
      int i = (k - 12) / 3, j = (k - 12) % 3;
      if (dproj) dproj[i * 4 + (j == 2 ? 3 : j)] = v;
    } else if (dcampos) {
      dcampos[k - 24] = v;
    }
  }
  if (k < 4) {  // matrix entries that never receive gradient
    if (dview) dview[k * 4 + 3] = 0.f;
    if (dproj) dproj[k * 4 + 2] = 0.f;
  }
}

void launch_preprocess_bwd(const CamDev& cam, int P, int M, int C, const float* means3D, const float* shs,
                           const float* colors, const float* opac, const float* scales, const float* rots,
                           const float* cov3d, const int32_t* radii, GeomView g, BinView b, size_t N_cap, BwdView bw,
                           float* dmeans3D, float* dmeans2D, float* dshs, float* dcolors, float* dopac, float* dscales,
                           float* drots, float* dcov3d, bool want_cam, int flags, hipStream_t s) {
  if (P <= 0) return;
  const bool vec = shs && (((uintptr_t)shs & 15) == 0) && (!dshs || ((uintptr_t)dshs & 15) == 0);
  auto kern = (vec && M == 16) ? preprocess_bwd_kernel<16> : ((vec && M == 4) ? preprocess_bwd_kernel<4> : preprocess_bwd_kernel<0>);
  hipLaunchKernelGGL(kern, dim3((P + PP_BLOCK - 1) / PP_BLOCK), dim3(PP_BLOCK), 0, s, cam, P, M, C,
                     means3D, shs, colors, opac, scales, rots, cov3d, radii, g, b,
                     (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap), bw.dsub, bw.campartial, dmeans3D,
                     dmeans2D, dshs, dcolors, dopac, dscales, drots, dcov3d, want_cam ? 1 : 0, flags);
}

void launch_camgrad_finish(BwdView bw, float* dview, float* dproj, float* dcampos, hipStream_t s) {
  hipLaunchKernelGGL(camgrad_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)bw.campartial, bw.nrows, dview,
                     dproj, dcampos);
}

__global__ void mark_visible_kernel(CamDev cam, int P, const float* __restrict__ means3D, uint8_t* __restrict__ vis) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float* V = cam.view;
  float tz = means3D[(size_t)idx * 3] * V[2] + means3D[(size_t)idx * 3 + 1] * V[6] + means3D[(size_t)idx * 3 + 2] * V[10] + V[14];
  vis[idx] = tz > 0.2f ? 1 : 0;
}
void launch_mark_visible(const CamDev& cam, int P, const float* means3D, uint8_t* visible, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, cam, P, means3D, visible);
}
