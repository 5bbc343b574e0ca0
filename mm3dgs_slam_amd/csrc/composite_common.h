// Pieces shared by the compositing kernels (composite.hip, composite_bwd2.hip): the skip / stop constants, the power of the
// Gaussian (ONE instruction sequence, so that forward and backward take identical skip decisions), the packed splat record.
#pragma once
#include "mm3dgs_common.h"

#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f

__device__ __forceinline__ int xcd_tile(int bid, int T, int mode) {
  if (mode == 0) return bid;
  // workgroup b runs on XCD b % 8 (observed placement; only speed depends on it): give each XCD a contiguous
  // span of tiles so neighbouring tiles, which share splats, hit the same 4 MB L2.
  int per = (T + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

// identical instruction sequence in forward and backward so both take the same skip decisions
__device__ __forceinline__ float splat_power(float dx, float dy, float ca, float cb, float cc) {
  return fmaf(-0.5f, fmaf(ca * dx, dx, cc * dy * dy), -cb * dx * dy);
}

struct SplatRec { float4 A, B, C; };  // A: px py conA conB | B: conC opacity c0 c1 | C: c2..c5

template <int C>
__device__ __forceinline__ SplatRec load_rec(const float* __restrict__ splat, uint32_t id, bool have) {
  SplatRec r;
  r.A = r.B = r.C = make_float4(0.f, 0.f, 0.f, 0.f);
  if (have) {
    const float4* sp = (const float4*)(splat + (size_t)id * SPLAT_F);
    r.A = sp[0];
    r.B = sp[1];
    if (C > 2) r.C = sp[2];
  }
  return r;
}

