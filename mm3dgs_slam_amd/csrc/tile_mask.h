// Which of the sixteen 4x4-pixel blocks of a tile list a splat: the block lists hold a splat only where the bound of its
// { alpha >= 1/255 } region can reach (sort_tile.h emits the lists, the backward gather has to know which gradient records of
// the splat's block rectangle were written).  The rule is evaluated in three kernels (binning, sort, gather) and they must
// agree bit for bit, so every operation that the compiler could contract into an fma is spelled as a rounded intrinsic.
//
//   alpha >= 1/255  <=>  d^T Q d <= 2 tau,  tau = ln(255 o),  Q = [[A.z, A.w], [A.w, B.x]]   (A, B = first 32 B of the splat)
// A block (pixel centres [x0, x0+3] x [y0, y0+3]) is listed iff the axis-aligned bound of that ellipse overlaps it AND it comes
// within sqrt(2 tau lambda_max) of the centre (exact for isotropic splats, where the box test alone keeps the corners a disc
// cannot reach).  Both are necessary conditions, so compositing the block lists equals compositing the tile list.
#pragma once
#include "mm3dgs_common.h"

struct MaskConsts {
  float cx, cy;      // splat centre (pixels)
  float hx, hy;      // half extents of the axis-aligned bound (with slack)
  float r2;          // squared reach of the disc bound (with slack)
  int mode;          // 0: nothing listed (alpha < 1/255 everywhere), 1: tests apply, 2: degenerate conic -> every block listed
};

__device__ __forceinline__ MaskConsts mask_consts(const float4 A, const float4 B) {
  MaskConsts m;
  m.cx = A.x; m.cy = A.y; m.hx = 0.f; m.hy = 0.f; m.r2 = 0.f;
  const float tau = __logf(__fmul_rn(255.f, B.y));
  const float det = __fsub_rn(__fmul_rn(A.z, B.x), __fmul_rn(A.w, A.w));
  if (!(det > 0.f)) { m.mode = 2; return m; }
  if (!(tau > 0.f)) { m.mode = 0; return m; }
  m.mode = 1;
  const float t2 = __fmul_rn(2.f, tau);
  const float k = __fdiv_rn(t2, det);
  m.hx = __fadd_rn(__fmul_rn(__fsqrt_rn(__fmul_rn(k, B.x)), 1.0002f), 0.002f);
  m.hy = __fadd_rn(__fmul_rn(__fsqrt_rn(__fmul_rn(k, A.z)), 1.0002f), 0.002f);
  const float sxx = __fdiv_rn(B.x, det), syy = __fdiv_rn(A.z, det), mid = __fmul_rn(0.5f, __fadd_rn(sxx, syy));
  const float lam = __fadd_rn(mid, __fsqrt_rn(fmaxf(__fsub_rn(__fmul_rn(mid, mid), __fdiv_rn(1.f, det)), 0.f)));
  m.r2 = __fadd_rn(__fmul_rn(__fmul_rn(t2, lam), 1.0004f), 0.01f);
  return m;
}

// is block (bx, by) (global block coordinates: pixel centres [4bx, 4bx+3] x [4by, 4by+3]) listed?  Tile-relative arithmetic,
// exactly as tile_block_mask evaluates it.
__device__ __forceinline__ bool block_listed(const MaskConsts& m, int bx, int by) {
  if (m.mode != 1) return m.mode == 2;
  const float cx = __fsub_rn(m.cx, (float)((bx >> 2) * TILE)), cy = __fsub_rn(m.cy, (float)((by >> 2) * TILE));
  const float lox = 4.f * (float)(bx & 3), hix = lox + 3.f, loy = 4.f * (float)(by & 3), hiy = loy + 3.f;   // exact small integers
  const bool inx = (__fsub_rn(cx, m.hx) <= hix) && (__fadd_rn(cx, m.hx) >= lox);
  const bool iny = (__fsub_rn(cy, m.hy) <= hiy) && (__fadd_rn(cy, m.hy) >= loy);
  const float dx = fmaxf(fmaxf(__fsub_rn(lox, cx), __fsub_rn(cx, hix)), 0.f), dy = fmaxf(fmaxf(__fsub_rn(loy, cy), __fsub_rn(cy, hiy)), 0.f);
  return inx && iny && (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) <= m.r2);
}

// 16-bit mask of tile (ttx, tty); bit L = 4 * (8x8 sub-tile) + (block in the sub-tile), the order of the block lists
__device__ __forceinline__ uint32_t tile_block_mask(const MaskConsts& m, int ttx, int tty) {
  if (m.mode != 1) return m.mode == 2 ? 0xffffu : 0u;
  const float cx = __fsub_rn(m.cx, (float)(ttx * TILE)), cy = __fsub_rn(m.cy, (float)(tty * TILE));
  bool bx[4], by[4];
  float ex[4], ey[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float lo = 4.f * q, hi = 4.f * q + 3.f;
    bx[q] = (__fsub_rn(cx, m.hx) <= hi) && (__fadd_rn(cx, m.hx) >= lo);
    by[q] = (__fsub_rn(cy, m.hy) <= hi) && (__fadd_rn(cy, m.hy) >= lo);
    const float dxq = fmaxf(fmaxf(__fsub_rn(lo, cx), __fsub_rn(cx, hi)), 0.f), dyq = fmaxf(fmaxf(__fsub_rn(lo, cy), __fsub_rn(cy, hi)), 0.f);
    ex[q] = __fmul_rn(dxq, dxq); ey[q] = __fmul_rn(dyq, dyq);
  }
  uint32_t mask = 0;
#pragma unroll
  for (int my = 0; my < 4; my++)
#pragma unroll
    for (int kx = 0; kx < 4; kx++) {
      const int L = 4 * ((my >> 1) * 2 + (kx >> 1)) + (my & 1) * 2 + (kx & 1);
      if (bx[kx] && by[my] && (__fadd_rn(ex[kx], ey[my]) <= m.r2)) mask |= 1u << L;
    }
  return mask;
}

// blocks outside the splat's block rectangle are never listed (it bounds the same region with more slack; belt and braces)
__device__ __forceinline__ uint32_t clip_mask_to_rect(uint32_t mask, int ttx, int tty, const BlkRect& br) {
#pragma unroll
  for (int L = 0; L < NLIST; L++) {
    const int bx = ttx * 4 + ((L >> 2) & 1) * 2 + (L & 1) - br.bx0, by = tty * 4 + (L >> 3) * 2 + ((L >> 1) & 1) - br.by0;
    if (bx < 0 || by < 0 || bx >= br.bw || by >= br.bh) mask &= ~(1u << L);
  }
  return mask;
}

// clip_mask_to_rect(tile_block_mask(m, ttx, tty), ttx, tty, br) with the rectangle test folded into the per-column / per-row predicates
// (8 integer compares instead of 64): the same mask, bit for bit -- a block is listed iff its column passes, its row passes and the
// disc reaches it, and "inside the block rectangle" is a column property AND a row property too.
__device__ __forceinline__ uint32_t tile_block_mask_in_rect(const MaskConsts& m, int ttx, int tty, const BlkRect& br) {
  if (m.mode == 0) return 0u;
  const float cx = __fsub_rn(m.cx, (float)(ttx * TILE)), cy = __fsub_rn(m.cy, (float)(tty * TILE));
  bool bx[4], by[4];
  float ex[4], ey[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    bx[q] = (unsigned)(ttx * 4 + q - br.bx0) < (unsigned)br.bw;
    by[q] = (unsigned)(tty * 4 + q - br.by0) < (unsigned)br.bh;
    ex[q] = 0.f; ey[q] = 0.f;
    if (m.mode == 1) {
      const float lo = 4.f * q, hi = 4.f * q + 3.f;
      bx[q] = bx[q] && (__fsub_rn(cx, m.hx) <= hi) && (__fadd_rn(cx, m.hx) >= lo);
      by[q] = by[q] && (__fsub_rn(cy, m.hy) <= hi) && (__fadd_rn(cy, m.hy) >= lo);
      const float dxq = fmaxf(fmaxf(__fsub_rn(lo, cx), __fsub_rn(cx, hi)), 0.f), dyq = fmaxf(fmaxf(__fsub_rn(lo, cy), __fsub_rn(cy, hi)), 0.f);
      ex[q] = __fmul_rn(dxq, dxq); ey[q] = __fmul_rn(dyq, dyq);
    }
  }
  uint32_t mask = 0;
#pragma unroll
  for (int my = 0; my < 4; my++)
#pragma unroll
    for (int kx = 0; kx < 4; kx++) {
      const int L = 4 * ((my >> 1) * 2 + (kx >> 1)) + (my & 1) * 2 + (kx & 1);
      if (bx[kx] && by[my] && (m.mode == 2 || __fadd_rn(ex[kx], ey[my]) <= m.r2)) mask |= 1u << L;
    }
  return mask;
}
