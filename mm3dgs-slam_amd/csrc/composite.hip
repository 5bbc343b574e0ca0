// Alpha compositing (forward) and its per-pixel reverse traversal (backward) for gfx950 (wave64).
//
// Work decomposition: one 256-lane workgroup per 16x16 tile; each of its 4 waves owns an 8x8 sub-tile.  The tile's
// depth-ordered splat list is staged through LDS 256 entries at a time (three float4 arrays, every lane later reads
// the same address = LDS broadcast).  While staging, each lane also computes the axis-aligned bound of the region
// where its splat can reach alpha >= 1/255 ( d^T Q d <= 2 ln(255 o) ) and tests it against the four sub-tiles; one
// 64-bit ballot per (64-entry group, sub-tile) is kept in LDS.  A wave then walks only the set bits of ITS masks with
// scalar bit-scan instructions, so splats that cannot touch its 64 pixels cost it nothing.  The bound is conservative
// (plus a small margin), so the result is exactly the reference rule applied to every (pixel, splat) pair:
//   alpha = min(0.99, o * exp(power)); skip power > 0 or alpha < 1/255; stop before the splat that would push T
//   below 1e-4; out = sum c alpha T + T_final * bg        (SURVEY.md Appendix A).
//
// Backward: per-lane gradient terms of one splat (2 mean + 3 conic + 1 opacity + C colour values) are reduced over
// the wave with a multi-value DPP butterfly (two halving levels on lane bits 0/1 with quad_perm, then row rotations:
// ~3 VALU ops per value instead of 6 per value), committed with ONE LDS atomic instruction per (wave, splat) into a
// per-batch LDS accumulator, and flushed with one global float atomic per (tile, splat, component).
#include "mm3dgs_common.h"

#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f

__device__ __forceinline__ int xcd_tile(int bid, int T) {
  // workgroup b runs on XCD b % 8 (observed placement; only speed depends on it): give each XCD a contiguous
  // span of tiles so neighbouring tiles, which share splats, hit the same 4 MB L2.
  int per = (T + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

// identical instruction sequence in forward and backward so both take the same skip decisions
__device__ __forceinline__ float splat_power(float dx, float dy, float ca, float cb, float cc) {
  return fmaf(-0.5f, fmaf(ca * dx, dx, cc * dy * dy), -cb * dx * dy);
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Stage entry `tid` of the current batch into LDS and publish the per-sub-tile visibility ballots.
template <int C>
__device__ __forceinline__ void stage_batch(bool have, uint32_t id, const float* __restrict__ splat, int tid, float4* sA,
                                            float4* sB, float4* sC, unsigned long long (*smask)[4], float tile_x0,
                                            float tile_y0) {
  bool ovx0 = false, ovx1 = false, ovy0 = false, ovy1 = false;
  if (have) {
    const float4* sp = (const float4*)(splat + (size_t)id * SPLAT_F);
    float4 A = sp[0], B = sp[1];
    sA[tid] = A;
    sB[tid] = B;
    if (C > 2) sC[tid] = sp[2];
    // alpha >= 1/255  <=>  d^T Q d <= 2 tau, tau = ln(255 o), Q = [[A.z, A.w],[A.w, B.x]]
    float tau = __logf(255.f * B.y);
    float det = A.z * B.x - A.w * A.w;
    if (det > 0.f) {
      if (tau > 0.f) {
        float k = 2.f * tau / det;
        float hx = sqrtf(k * B.x) * 1.0002f + 0.002f;
        float hy = sqrtf(k * A.z) * 1.0002f + 0.002f;
        float xl = A.x - hx - tile_x0, xh = A.x + hx - tile_x0;
        float yl = A.y - hy - tile_y0, yh = A.y + hy - tile_y0;
        ovx0 = (xl <= 7.f) && (xh >= 0.f);
        ovx1 = (xl <= 15.f) && (xh >= 8.f);
        ovy0 = (yl <= 7.f) && (yh >= 0.f);
        ovy1 = (yl <= 15.f) && (yh >= 8.f);
      }
    } else {
      ovx0 = ovx1 = ovy0 = ovy1 = true;  // degenerate conic: no culling, the exact per-pixel rule decides
    }
  }
  unsigned long long m0 = __ballot(ovx0 && ovy0), m1 = __ballot(ovx1 && ovy0);
  unsigned long long m2 = __ballot(ovx0 && ovy1), m3 = __ballot(ovx1 && ovy1);
  if ((tid & 63) == 0) {
    int g = tid >> 6;
    smask[g][0] = m0; smask[g][1] = m1; smask[g][2] = m2; smask[g][3] = m3;
  }
}

template <int C>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, float* __restrict__ out) {
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx0 = (tile % cam.gx) * TILE, ty0 = (tile / cam.gx) * TILE;
  const int px = tx0 + (wv & 1) * 8 + (lane & 7);
  const int py = ty0 + (wv >> 1) * 8 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t start = min(iv.ranges[tile], N_cap), end = min(iv.ranges[tile + 1], N_cap);

  __shared__ float4 sA[256];  // px, py, conA, conB
  __shared__ float4 sB[256];  // conC, opacity, c0, c1
  __shared__ float4 sC[256];  // c2..c5
  __shared__ unsigned long long smask[4][4];

  float Tr = 1.f;
  float acc[C];
#pragma unroll
  for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
  uint32_t last_contributor = 0;
  bool done = !inside;

  for (uint32_t base = start; base < end; base += 256) {
    if (__syncthreads_count(done) == 256) break;
    const uint32_t k = base + tid;
    const bool have = k < end;
    const uint32_t id = have ? b.point_list[k] : 0u;
    stage_batch<C>(have, id, g.splat, tid, sA, sB, sC, smask, (float)tx0, (float)ty0);
    __syncthreads();
    const uint32_t pos0 = base - start + 1;  // 1-based list position of entry 0 of this batch
    for (int grp = 0; grp < 4; grp++) {
      unsigned long long m = uniform_u64(smask[grp][wv]);
      if (m == 0ull) continue;
      if (__ballot(!done) == 0ull) break;
      int j = grp * 64 + __builtin_ctzll(m);
      m &= m - 1;
      float4 A = sA[j], B = sB[j], Cc = sC[C > 2 ? j : 0];
      while (true) {
        // prefetch the next visible splat's record while this one is evaluated
        const int jn = m ? grp * 64 + __builtin_ctzll(m) : j;
        const float4 nA = sA[jn], nB = sB[jn], nC = sC[C > 2 ? jn : 0];
        const float dx = A.x - pxf, dy = A.y - pyf;
        const float power = splat_power(dx, dy, A.z, A.w, B.x);
        const float alpha = fminf(0.99f, B.y * __expf(power));
        const bool ok = !done && !(power > 0.f) && !(alpha < ALPHA_MIN);
        const float test_T = Tr * (1.f - alpha);
        const bool stop = ok && (test_T < T_EPS);
        const bool contrib = ok && !stop;
        done = done || stop;
        const float w = contrib ? alpha * Tr : 0.f;
        if (C > 0) acc[0] = fmaf(B.z, w, acc[0]);
        if (C > 1) acc[1] = fmaf(B.w, w, acc[1]);
        if (C > 2) acc[2] = fmaf(Cc.x, w, acc[2]);
        if (C > 3) acc[3] = fmaf(Cc.y, w, acc[3]);
        if (C > 4) acc[4] = fmaf(Cc.z, w, acc[4]);
        if (C > 5) acc[5] = fmaf(Cc.w, w, acc[5]);
        Tr = contrib ? test_T : Tr;
        last_contributor = contrib ? pos0 + (uint32_t)j : last_contributor;
        if (m == 0ull) break;
        if (__ballot(!done) == 0ull) break;
        m &= m - 1;
        j = jn; A = nA; B = nB; Cc = nC;
      }
    }
  }
  if (inside) {
    size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
    iv.final_T[pix] = Tr;
    iv.n_contrib[pix] = last_contributor;
#pragma unroll
    for (int ch = 0; ch < C; ch++) out[ch * HW + pix] = acc[ch] + (ch < 3 ? Tr * cam.bg[ch] : 0.f);
  }
}

// ---- multi-value wave reduction ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_all(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define QP_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define QP_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define ROW_ROR4 0x124
#define ROW_ROR8 0x128

// Sum NV per-lane values over the 64 lanes.  On return lane l holds, in the returned float, the total over its
// 16-lane row of value index  slot(l) = q + M2*b1 + M1*b0  (b0,b1 = lane bits 0,1; q = (l>>2)&3), valid when
// commit_lane<NV>(l) is true; the four rows are merged by the caller's LDS atomic.
template <int NV>
struct WaveReduce {
  static constexpr int M1 = (NV + 1) / 2;
  static constexpr int M2 = (M1 + 1) / 2;
  static_assert(NV >= 1 && NV <= 16, "NV out of range");
  __device__ static __forceinline__ int slot(int lane) {
    int b0 = lane & 1, b1 = (lane >> 1) & 1, q = (lane >> 2) & 3;
    int i1 = q + M2 * b1;
    int idx = i1 + M1 * b0;
    bool ok = (q < M2) && (i1 < M1) && (idx < NV);
    return ok ? idx : -1;
  }
  __device__ static __forceinline__ float run(const float (&v)[NV], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float l1[M1];
#pragma unroll
    for (int i = 0; i < M1; i++) {
      float sa = v[i] + dpp_all<QP_XOR1>(v[i]);
      if (i + M1 < NV) {
        float sb = v[i + M1] + dpp_all<QP_XOR1>(v[i + M1]);
        l1[i] = b0 ? sb : sa;
      } else {
        l1[i] = sa;
      }
    }
    float l2[M2];
#pragma unroll
    for (int i = 0; i < M2; i++) {
      float sa = l1[i] + dpp_all<QP_XOR2>(l1[i]);
      if (i + M2 < M1) {
        float sb = l1[i + M2] + dpp_all<QP_XOR2>(l1[i + M2]);
        l2[i] = b1 ? sb : sa;
      } else {
        l2[i] = sa;
      }
    }
    // sum the four quads of each row (cyclic rotations: every lane ends with its residue-class total)
#pragma unroll
    for (int i = 0; i < M2; i++) {
      l2[i] += dpp_all<ROW_ROR4>(l2[i]);
      l2[i] += dpp_all<ROW_ROR8>(l2[i]);
    }
    const int q = (lane >> 2) & 3;
    float r = l2[0];
#pragma unroll
    for (int i = 1; i < M2; i++) r = (q == i) ? l2[i] : r;
    return r;
  }
};

template <int C>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, const float* __restrict__ dL_dout,
                     float* __restrict__ dsplat) {
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx0 = (tile % cam.gx) * TILE, ty0 = (tile / cam.gx) * TILE;
  const int px = tx0 + (wv & 1) * 8 + (lane & 7);
  const int py = ty0 + (wv >> 1) * 8 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t start = min(iv.ranges[tile], N_cap), end = min(iv.ranges[tile + 1], N_cap);
  if (end == start) return;

  constexpr int NV = 6 + C;
  __shared__ float4 sA[256];
  __shared__ float4 sB[256];
  __shared__ float4 sC[256];
  __shared__ uint32_t sid[256];
  __shared__ __attribute__((aligned(16))) float sacc[256][SPLAT_F];  // per-batch gradient accumulators (NV used)
  __shared__ unsigned long long smask[4][4];
  __shared__ uint32_t smax;

  const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
  const float T_final = inside ? iv.final_T[pix] : 0.f;
  const uint32_t last_contributor = inside ? iv.n_contrib[pix] : 0u;
  float dL[C];
  float bg_dot = 0.f;
#pragma unroll
  for (int ch = 0; ch < C; ch++) {
    dL[ch] = inside ? dL_dout[ch * HW + pix] : 0.f;
    if (ch < 3) bg_dot += cam.bg[ch] * dL[ch];
  }
  const float Tf_bg = T_final * bg_dot;
  float Tr = T_final;
  float behind[C];  // colour accumulated behind the current list position
#pragma unroll
  for (int ch = 0; ch < C; ch++) behind[ch] = 0.f;

  // nothing behind the deepest contributor of any pixel of the tile matters
  if (tid == 0) smax = 0;
  __syncthreads();
  atomicMax(&smax, last_contributor);
  __syncthreads();
  const uint32_t todo = smax;
  if (todo == 0) return;

  const int my_slot = WaveReduce<NV>::slot(lane);

  for (uint32_t base = 0; base < todo; base += 256) {
    __syncthreads();  // previous batch fully consumed / flushed
    const uint32_t k = base + tid;
    const bool have = k < todo;
    const uint32_t id = have ? b.point_list[start + (todo - 1 - k)] : 0u;
    sid[tid] = id;
    stage_batch<C>(have, id, g.splat, tid, sA, sB, sC, smask, (float)tx0, (float)ty0);
    {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float4* row = (float4*)&sacc[tid][0];
      row[0] = z; row[1] = z; row[2] = z;
    }
    __syncthreads();
    for (int grp = 0; grp < 4; grp++) {
      unsigned long long m = uniform_u64(smask[grp][wv]);
      if (m == 0ull) continue;
      int j = grp * 64 + __builtin_ctzll(m);
      m &= m - 1;
      float4 A = sA[j], B = sB[j], Cc = sC[C > 2 ? j : 0];
      while (true) {
        const int jn = m ? grp * 64 + __builtin_ctzll(m) : j;
        const float4 nA = sA[jn], nB = sB[jn], nC = sC[C > 2 ? jn : 0];
        const uint32_t pos = todo - 1 - (base + (uint32_t)j);  // 0-based index in the tile list
        const float dx = A.x - pxf, dy = A.y - pyf;
        const float power = splat_power(dx, dy, A.z, A.w, B.x);
        const float G = __expf(power);
        const float alpha = fminf(0.99f, B.y * G);
        const bool valid = (pos < last_contributor) && !(power > 0.f) && !(alpha < ALPHA_MIN);
        if (__ballot(valid) != 0ull) {
          const float a_eff = valid ? alpha : 0.f;
          const float G_eff = valid ? G : 0.f;
          const float r = __builtin_amdgcn_rcpf(1.f - a_eff);
          Tr *= r;  // transmittance in front of this splat
          const float w = a_eff * Tr;
          float col[C];
          if (C > 0) col[0] = B.z;
          if (C > 1) col[1] = B.w;
          if (C > 2) col[2] = Cc.x;
          if (C > 3) col[3] = Cc.y;
          if (C > 4) col[4] = Cc.z;
          if (C > 5) col[5] = Cc.w;
          float vals[NV];
          float dLa = 0.f;
#pragma unroll
          for (int ch = 0; ch < C; ch++) {
            const float diff = col[ch] - behind[ch];
            dLa = fmaf(diff, dL[ch], dLa);
            behind[ch] = fmaf(a_eff, diff, behind[ch]);
            vals[6 + ch] = w * dL[ch];
          }
          dLa = dLa * Tr - Tf_bg * r;
          const float dL_dG = B.y * dLa;
          const float gdx = G_eff * dx, gdy = G_eff * dy;
          vals[0] = -dL_dG * (gdx * A.z + gdy * A.w);
          vals[1] = -dL_dG * (gdy * B.x + gdx * A.w);
          vals[2] = -0.5f * gdx * dx * dL_dG;
          vals[3] = -gdx * dy * dL_dG;
          vals[4] = -0.5f * gdy * dy * dL_dG;
          vals[5] = G_eff * dLa;
          const float tot = WaveReduce<NV>::run(vals, lane);
          if (my_slot >= 0) atomicAdd(&sacc[j][my_slot], tot);
        }
        if (m == 0ull) break;
        m &= m - 1;
        j = jn; A = nA; B = nB; Cc = nC;
      }
    }
    __syncthreads();
    if (have) {
      const float4* row = (const float4*)&sacc[tid][0];
      float4 r0 = row[0], r1 = row[1], r2 = row[2];
      float s[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      uint32_t any = 0;
#pragma unroll
      for (int v = 0; v < NV; v++) any |= __float_as_uint(s[v]);
      if (any << 1) {  // something other than +-0 was accumulated
        float* dst = dsplat + (size_t)id * SPLAT_F;
#pragma unroll
        for (int v = 0; v < NV; v++) atomicAdd(&dst[v], s[v]);
      }
    }
  }
}

template <int C>
static void launch_fwd_c(const CamDev& cam, GeomView g, ImageView iv, BinView b, uint32_t ncap, float* out, hipStream_t s) {
  int T = cam.gx * cam.gy;
  int grid = ((T + 7) / 8) * 8;
  hipLaunchKernelGGL((composite_fwd_kernel<C>), dim3(grid), dim3(256), 0, s, cam, g, iv, b, ncap, out);
}
template <int C>
static void launch_bwd_c(const CamDev& cam, GeomView g, ImageView iv, BinView b, uint32_t ncap, const float* dL, float* dsplat,
                         hipStream_t s) {
  int T = cam.gx * cam.gy;
  int grid = ((T + 7) / 8) * 8;
  hipLaunchKernelGGL((composite_bwd_kernel<C>), dim3(grid), dim3(256), 0, s, cam, g, iv, b, ncap, dL, dsplat);
}

void launch_composite_fwd(const CamDev& cam, int C, GeomView g, ImageView iv, BinView b, size_t N_cap, float* out,
                          hipStream_t s) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  switch (C) {
    case 1: launch_fwd_c<1>(cam, g, iv, b, ncap, out, s); break;
    case 2: launch_fwd_c<2>(cam, g, iv, b, ncap, out, s); break;
    case 3: launch_fwd_c<3>(cam, g, iv, b, ncap, out, s); break;
    case 4: launch_fwd_c<4>(cam, g, iv, b, ncap, out, s); break;
    case 5: launch_fwd_c<5>(cam, g, iv, b, ncap, out, s); break;
    default: launch_fwd_c<6>(cam, g, iv, b, ncap, out, s); break;
  }
}
void launch_composite_bwd(const CamDev& cam, int C, GeomView g, ImageView iv, BinView b, size_t N_cap, const float* dL,
                          float* dsplat, hipStream_t s) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  switch (C) {
    case 1: launch_bwd_c<1>(cam, g, iv, b, ncap, dL, dsplat, s); break;
    case 2: launch_bwd_c<2>(cam, g, iv, b, ncap, dL, dsplat, s); break;
    case 3: launch_bwd_c<3>(cam, g, iv, b, ncap, dL, dsplat, s); break;
    case 4: launch_bwd_c<4>(cam, g, iv, b, ncap, dL, dsplat, s); break;
    case 5: launch_bwd_c<5>(cam, g, iv, b, ncap, dL, dsplat, s); break;
    default: launch_bwd_c<6>(cam, g, iv, b, ncap, dL, dsplat, s); break;
  }
}
