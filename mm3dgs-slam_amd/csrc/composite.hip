// Alpha compositing (forward) and its per-pixel reverse traversal (backward) for gfx950.
// One 256-lane workgroup per 16x16 tile; each of its 4 waves owns an 8x8 sub-tile (compact footprint, so a small
// splat usually touches one or two waves and the other waves skip it with a wave-uniform branch).
// The tile's depth-ordered list is staged through LDS 256 splats at a time as three float4 arrays; inside the loop
// every lane reads the same LDS address (broadcast, conflict-free).
// Backward: per-lane gradient terms are summed across the wave with DPP (no LDS), the wave totals are merged into a
// per-batch LDS accumulator, and one global float atomic per (Gaussian, tile, component) flushes it -- instead of
// one atomic per (Gaussian, pixel, component).
//
// Compositing rule: alpha = min(0.99, o * exp(power)), skip power > 0 or alpha < 1/255,
// stop before the splat that would push T below 1e-4, out = sum c alpha T + T_final * bg.
#include "mm3dgs_common.h"

#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f

__device__ __forceinline__ int xcd_tile(int bid, int T) {
  // workgroup b runs on XCD b % 8 (observed placement; only speed depends on it): give each XCD a contiguous
  // span of tiles so neighbouring tiles, which share splats, hit the same 4 MB L2.
  int per = (T + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

template <int C>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, float* __restrict__ out) {
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int px = (tile % cam.gx) * TILE + (wv & 1) * 8 + (lane & 7);
  const int py = (tile / cam.gx) * TILE + (wv >> 1) * 8 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t start = min(iv.ranges[tile], N_cap), end = min(iv.ranges[tile + 1], N_cap);

  __shared__ float4 sA[256];  // px, py, conA, conB
  __shared__ float4 sB[256];  // conC, opacity, c0, c1
  __shared__ float4 sC[256];  // c2..c5

  float Tr = 1.f;
  float acc[C];
#pragma unroll
  for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
  uint32_t contributor = 0, last_contributor = 0;
  bool done = !inside;

  for (uint32_t base = start; base < end; base += 256) {
    if (__syncthreads_count(done) == 256) break;
    uint32_t k = base + tid;
    if (k < end) {
      uint32_t id = b.point_list[k];
      const float4* sp = (const float4*)(g.splat + (size_t)id * SPLAT_F);
      sA[tid] = sp[0];
      sB[tid] = sp[1];
      if (C > 2) sC[tid] = sp[2];
    }
    __syncthreads();
    const int cnt = (int)min(256u, end - base);
    for (int j = 0; !done && j < cnt; j++) {
      contributor++;
      float4 A = sA[j];
      float4 B = sB[j];
      float dx = A.x - pxf, dy = A.y - pyf;
      float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
      if (power > 0.f) continue;
      float alpha = fminf(0.99f, B.y * __expf(power));
      if (alpha < ALPHA_MIN) continue;
      float test_T = Tr * (1.f - alpha);
      if (test_T < T_EPS) { done = true; continue; }
      float w = alpha * Tr;
      if (C > 0) acc[0] += B.z * w;
      if (C > 1) acc[1] += B.w * w;
      if (C > 2) {
        float4 Cc = sC[j];
        acc[2] += Cc.x * w;
        if (C > 3) acc[3] += Cc.y * w;
        if (C > 4) acc[4] += Cc.z * w;
        if (C > 5) acc[5] += Cc.w * w;
      }
      Tr = test_T;
      last_contributor = contributor;
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

template <int C>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, const float* __restrict__ dL_dout,
                     float* __restrict__ dsplat) {
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int px = (tile % cam.gx) * TILE + (wv & 1) * 8 + (lane & 7);
  const int py = (tile / cam.gx) * TILE + (wv >> 1) * 8 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t start = min(iv.ranges[tile], N_cap), end = min(iv.ranges[tile + 1], N_cap);
  if (end == start) return;

  __shared__ float4 sA[256];
  __shared__ float4 sB[256];
  __shared__ float4 sC[256];
  __shared__ uint32_t sid[256];
  __shared__ float sacc[256][SPLAT_F];  // per-batch gradient accumulators (6 + C used)
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
  float Tr = T_final;
  float accum_rec[C], last_color[C];
#pragma unroll
  for (int ch = 0; ch < C; ch++) { accum_rec[ch] = 0.f; last_color[ch] = 0.f; }
  float last_alpha = 0.f;

  // nothing behind the deepest contributor of any pixel of the tile matters
  if (tid == 0) smax = 0;
  __syncthreads();
  atomicMax(&smax, last_contributor);
  __syncthreads();
  const uint32_t todo = smax;
  if (todo == 0) return;

  constexpr int NV = 6 + C;
  for (uint32_t base = 0; base < todo; base += 256) {
    __syncthreads();  // previous batch fully consumed / flushed
    uint32_t k = base + tid;
    if (k < todo) {
      uint32_t id = b.point_list[start + (todo - 1 - k)];
      sid[tid] = id;
      const float4* sp = (const float4*)(g.splat + (size_t)id * SPLAT_F);
      sA[tid] = sp[0];
      sB[tid] = sp[1];
      if (C > 2) sC[tid] = sp[2];
    }
#pragma unroll
    for (int v = 0; v < NV; v++) sacc[tid][v] = 0.f;
    __syncthreads();
    const int cnt = (int)min(256u, todo - base);
    for (int j = 0; j < cnt; j++) {
      const uint32_t pos = todo - 1 - (base + j);  // index in the tile list
      float4 A = sA[j];
      float4 B = sB[j];
      float dx = A.x - pxf, dy = A.y - pyf;
      float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
      float G = __expf(power);
      float alpha = fminf(0.99f, B.y * G);
      bool valid = (pos < last_contributor) && !(power > 0.f) && !(alpha < ALPHA_MIN);
      if (__ballot(valid) == 0ull) continue;  // wave-uniform: this 8x8 sub-tile does not see the splat
      float col[C];
      if (C > 0) col[0] = B.z;
      if (C > 1) col[1] = B.w;
      if (C > 2) {
        float4 Cc = sC[j];
        col[2] = Cc.x;
        if (C > 3) col[3] = Cc.y;
        if (C > 4) col[4] = Cc.z;
        if (C > 5) col[5] = Cc.w;
      }
      float vals[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) vals[v] = 0.f;
      if (valid) {
        Tr = Tr / (1.f - alpha);
        const float w = alpha * Tr;
        float dL_dalpha = 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
          accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
          last_color[ch] = col[ch];
          dL_dalpha += (col[ch] - accum_rec[ch]) * dL[ch];
          vals[6 + ch] = w * dL[ch];
        }
        dL_dalpha *= Tr;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        const float dL_dG = B.y * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        vals[0] = dL_dG * (-gdx * A.z - gdy * A.w);
        vals[1] = dL_dG * (-gdy * B.x - gdx * A.w);
        vals[2] = -0.5f * gdx * dx * dL_dG;
        vals[3] = -gdx * dy * dL_dG;
        vals[4] = -0.5f * gdy * dy * dL_dG;
        vals[5] = G * dL_dalpha;
      }
#pragma unroll
      for (int v = 0; v < NV; v++) {
        float s = wave_sum_to_lane63(vals[v]);
        if (lane == 63) atomicAdd(&sacc[j][v], s);
      }
    }
    __syncthreads();
    if (tid < cnt) {
      float* dst = dsplat + (size_t)sid[tid] * SPLAT_F;
#pragma unroll
      for (int v = 0; v < NV; v++) {
        float s = sacc[tid][v];
        if (s != 0.f) atomicAdd(&dst[v], s);
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
