// Pieces shared by the compositing kernels (composite.hip): the skip / stop constants, the power of the
// Gaussian (ONE instruction sequence, so that forward and backward take identical skip decisions), the packed splat record.
#pragma once
#include "mm3dgs_common.h"

// Gradient records of the SLAM modes are packed at their real size: 10 floats (40 B) per (block, splat) in mapping, 7 floats
// (28 B) in tracking -- round 1 wrote 48 / 32 B (the backward compositor's HBM traffic was 3.4x its algorithmic bytes).  Records
// are then only 4-byte aligned: the wide accesses go through 4-byte-aligned vector types (global_load / store_dwordx4 need dword
// alignment only), and a reader that fetches whole float4s past a record's end gets the head of the next record in lanes it ignores.
// (Round 5 re-measured the aligned alternative, 48 / 32-byte strides: mapping backward 59.7 us against 58.0, backward projection 31.1 against 30.7 --
// the bytes cost more than the line straddles, profiles/r05_ab_rec48.txt.)
#define REC_MAP_F 10
#define REC_TRACK_F 7
// generic path: 6 + C floats, packed.  (A 48-byte stride -- 16-byte aligned records -- was measured in round 5 on the 1080p / 3 M pass: the backward
// compositor's WRITE_SIZE is unchanged, 2.34 GB -- every record leaves the L2 as one 64-byte write either way, 37 M records x 64 B -- and the launch
// takes 1551 us instead of 1276: the consumer reads a third more bytes.  profiles/r05_c5_rec12.txt)
#define GENERIC_RECF(C) (6 + (C))
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float4 ld4u(const float* p) { const f4u v = *(const f4u*)p; return make_float4(v.x, v.y, v.z, v.w); }
// zero the NV floats of a record
template <int NV>
__device__ __forceinline__ void zero_record(float* r) {
  const f4u z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f + 4 <= NV; f += 4) *(f4u*)(r + f) = z4;
  if ((NV & 3) >= 2) { const f2u z2 = {0.f, 0.f}; *(f2u*)(r + (NV & ~3)) = z2; }
  if (NV & 1) r[NV - 1] = 0.f;
}
// store the first `n` (1..4) floats of v at p
__device__ __forceinline__ void st_part(float* p, const float4& v, int n) {
  if (n >= 4) { const f4u t = {v.x, v.y, v.z, v.w}; *(f4u*)p = t; }
  else if (n == 3) { const f2u t = {v.x, v.y}; *(f2u*)p = t; p[2] = v.z; }
  else if (n == 2) { const f2u t = {v.x, v.y}; *(f2u*)p = t; }
  else if (n == 1) p[0] = v.x;
}

// exp of the Gaussian's power: v_exp_f32 on power * log2(e) (forward and backward use the same sequence: same decisions).
// (the correctly rounded expf changes none of the parity figures: measured in round 3)
#define SPLAT_EXP(p) __expf(p)
#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 0.0001f

__device__ __forceinline__ int xcd_tile(int bid, int T, int mode) {
  if (mode == 0) return bid;
  // workgroup b runs on XCD b % 8 (observed placement; only speed depends on it): give each XCD a contiguous
  // span of tiles so neighbouring tiles, which share splats, hit the same 4 MB L2.
  int per = (T + 7) >> 3;
  return (bid & 7) * per + (bid >> 3);
}

// SLAM kernels over persistent state: the load-balanced workgroup -> tile table of binning.hip's tile_order_kernel when it is valid
// for this grid (one scalar load; any permutation of the tiles is correct, only the speed depends on it)
__device__ __forceinline__ int slam_tile(const CamDev& cam, const ImageView& iv, int bid, int T) {
  // (the launch may hold more workgroups than tiles: slam_grid() -- those beyond the arithmetic map's range leave)
  int tile = bid < ((T + 7) >> 3) * 8 ? xcd_tile(bid, T, cam.tilemap) : T;
  if (cam.tile_table && iv.hdr->tile_order_tiles == tile_order_key(cam.H, cam.W)) {
    const uint32_t o = iv.tile_order[bid];
    tile = o ? (int)o - 1 : T;       // (0: a workgroup slot beyond its XCD's span)
  }
  return tile;
}
// workgroups of a SLAM compositor launch: ceil(T / 8) slots per XCD
static inline int slam_grid(const CamDev& cam, int T) {
  (void)cam;
  return ((T + 7) / 8) * 8;
}

// identical instruction sequence in forward and backward so both take the same skip decisions
__device__ __forceinline__ float splat_power(float dx, float dy, float ca, float cb, float cc) {
  return fmaf(-0.5f, fmaf(ca * dx, dx, cc * dy * dy), -cb * dx * dy);
}

// Issue priority of a compositing wave by the length of its walk (round 5).  The five waves of a SIMD share its vector pipe round-robin, so a wave with an
// unusually long list ends up running ALONE after its neighbours have left -- at half the pipe's rate: a lone wave cannot issue back to back
// (tools/ubench/valu_rate.hip: 2.7 ns per instruction with one wave per SIMD, 1.35 with four).  With a higher priority it runs at its own pace from the start
// and the short waves fill the slots it leaves.  The reference length is the mean walk of the render the workgroup -> tile table was built from
// (Mm3dgsHeader.mean_wave_steps, written by tile_order_kernel; 0 = unknown: no priorities -- the generic entry points).  A scheduling hint only: results are
// unaffected.  Thresholds 1.125 / 1.375 / 1.75 x the mean, measured against 1.06 / 1.25 / 1.5, 1.25 / 1.5 / 2 and 1.5 / 2 / 3 and against absolute step
// counts (profiles/r05_ab_wave_priority.txt): fused tracking kernel 67.2 -> 64.2 us, mapping backward 57.9 -> 56.8, sort + forward 32.7 -> 31.5; the hand-held
// sweep, whose newly seeded side carries tile lists of twice the mean, 24.6 -> 25.9 frames/s.
__device__ __forceinline__ void wave_prio_by_steps(uint32_t steps, uint32_t mean_steps) {
  if (mean_steps == 0u) return;
  if (16u * steps >= 28u * mean_steps) __builtin_amdgcn_s_setprio(3);
  else if (16u * steps >= 22u * mean_steps) __builtin_amdgcn_s_setprio(2);
  else if (16u * steps >= 18u * mean_steps) __builtin_amdgcn_s_setprio(1);
}
__device__ __forceinline__ void wave_prio_reset() { __builtin_amdgcn_s_setprio(0); }
// the reference length, honoured only while image_state's workgroup -> tile table is valid for this image (the word is written with the table:
// a state buffer that never saw tile_order_kernel, a generic entry point, another image size leave it stale -- ADVICE round 5); 0 = no priorities
__device__ __forceinline__ uint32_t wave_mean_steps(const CamDev& cam, const ImageView& iv) {
  return (cam.tile_table && iv.hdr->tile_order_tiles == tile_order_key(cam.H, cam.W)) ? iv.hdr->mean_wave_steps : 0u;
}
// (Re-deciding the priority every chunk from what is LEFT of the walk against what is left of a mean walk -- longest-remaining-first -- measured no better:
// sort + forward 31.9 against 31.5 us, the other two unchanged, the hand-held sweep the same; round 5.)

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

