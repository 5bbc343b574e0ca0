// Per-tile depth sort + emission of the sixteen 4x4-block lists, as a device function shared by sort_tiles_kernel
// (binning.hip) and the fused sort + forward-composite kernel (composite.hip).
#pragma once
#include "mm3dgs_common.h"

struct SortShared {            // LDS of one sorting workgroup besides the key array
  uint32_t wcnt[4][NLIST];     // per-wave entry counts of a 256-entry chunk, per block list
  uint32_t pre[4][NLIST];      // write cursor of (wave, list) for the chunk
  uint32_t run[NLIST];         // entries emitted so far per list (the final list lengths)
};

// ---- 4. per-tile sort -------------------------------------------------------------------------------------------
// All-ascending bitonic network ("flip" first sub-step, then half-cleaners): with every comparator pointing the
// same way, slots >= len behave as +inf padding that never moves, so arbitrary lengths need no real padding.
template <typename KeyAt>
__device__ __forceinline__ void bitonic_any_len(KeyAt&& at, int len, int tid, int nthreads) {
  int n = 1;
  while (n < len) n <<= 1;
  for (int k = 2; k <= n; k <<= 1) {
    // flip step: i in the lower half of its k-block pairs with the mirrored element of the upper half
    for (int p = tid; p < n / 2; p += nthreads) {
      int blk = p / (k / 2), off = p % (k / 2);
      int i = blk * k + off, j = blk * k + (k - 1 - off);
      if (j < len) {
        unsigned long long a = at(i), c = at(j);
        if (a > c) { at(i) = c; at(j) = a; }
      }
    }
    __syncthreads();
    for (int jdist = k / 4; jdist > 0; jdist >>= 1) {
      for (int p = tid; p < n / 2; p += nthreads) {
        int i = (p / jdist) * (2 * jdist) + (p % jdist), j = i + jdist;
        if (j < len) {
          unsigned long long a = at(i), c = at(j);
          if (a > c) { at(i) = c; at(j) = a; }
        }
      }
      __syncthreads();
    }
  }
}

#define RANK_SORT_MAX 1024  // lists up to this length are rank-sorted (needs 2 * RANK_SORT_MAX <= CAP keys of LDS)

// Handles a tile with lo < len <= CAP in LDS (sk[CAP]); when GLOBAL_TAIL it also sorts len > CAP in place in global
// memory (rare: > 16 K splats on one tile).  After sorting it emits the sixteen block lists and their lengths
// (iv.subcount and sh.run).  Returns false when this tier leaves the tile to another launch (nothing written).
// Every lane of the 256-lane workgroup must call it (barriers inside).
template <int CAP, bool GLOBAL_TAIL>
__device__ __forceinline__ bool sort_tile_body(int tile, int gx, int lo, const GeomView& g, const ImageView& iv, const BinView& b,
                                               uint32_t N_cap, int clean, unsigned long long* sk, SortShared& sh) {
  uint32_t (*wcnt)[NLIST] = sh.wcnt;
  uint32_t (*pre)[NLIST] = sh.pre;
  uint32_t* run = sh.run;
  const uint32_t start = min(iv.ranges[tile], N_cap), end = min(iv.ranges[tile + 1], N_cap);
  const int len = (int)(end - start);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (clean && tid == 0) { iv.tile_count[tile] = 0; iv.cursor[tile] = 0; }   // scatter_scan_kernel's counters stay zero
  if (lo == 0 && len == 0) {
    if (tid < NLIST) { iv.subcount[NLIST * tile + tid] = 0; run[tid] = 0; }
    return true;
  }
  if (len <= lo) return false;
  if (!GLOBAL_TAIL && len > CAP) return false;
  unsigned long long* gk = b.keys + start;
  const bool in_lds = len <= CAP;
  if (in_lds && len <= RANK_SORT_MAX) {
    // Run sort + rank merge (keys are unique).  (1) every wave bitonic-sorts 64-key runs in registers (21 compare-exchange
    // steps through the LDS crossbar, no barriers); (2) the rank of a key = its position in its own run + its lower bound
    // in every other run (7-step binary searches, four runs interleaved), and the key goes straight to its final slot.
    // ~250 instructions per key-lane instead of 3 x len for the all-pairs rank sort (870 at the SLAM average of 290).
    unsigned long long* sk2 = sk + RANK_SORT_MAX;
    const int nruns = (len + 63) >> 6;
    for (int r = wv; r < nruns; r += 4) {
      const int i = r * 64 + lane;
      unsigned long long key = i < len ? gk[i] : ~0ull;   // padding sorts to the end of the last run
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const unsigned long long other = __shfl_xor(key, j, 64);
          const bool up = (lane & k) == 0, lower = (lane & j) == 0;
          const bool take_min = lower == up;
          const bool other_less = other < key;
          key = (take_min == other_less) ? other : key;
        }
      }
      sk2[i] = key;
    }
    __syncthreads();
    for (int i = tid; i < len; i += 256) {
      const unsigned long long mine = sk2[i];
      const int own = i >> 6;
      int rank = i & 63;
      for (int r0 = 0; r0 < nruns; r0 += 4) {
        int pos[4] = {0, 0, 0, 0};
        const unsigned long long* run_base[4];
#pragma unroll
        for (int u = 0; u < 4; u++) run_base[u] = sk2 + (r0 + u < nruns ? r0 + u : own) * 64;   // absent run: harmless re-read
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) {
#pragma unroll
          for (int u = 0; u < 4; u++) pos[u] += (run_base[u][pos[u] + st - 1] < mine) ? st : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          pos[u] += (run_base[u][pos[u]] < mine) ? 1 : 0;
          rank += (r0 + u < nruns && r0 + u != own) ? pos[u] : 0;
        }
      }
      sk[rank] = mine;
    }
    __syncthreads();
  } else if (in_lds) {
    for (int i = tid; i < len; i += 256) sk[i] = gk[i];
    __syncthreads();
    if (len > 1) bitonic_any_len([&](int i) -> unsigned long long& { return sk[i]; }, len, tid, 256);
  } else {
    __syncthreads();
    bitonic_any_len([&](int i) -> unsigned long long& { return gk[i]; }, len, tid, 256);
  }
  // ---- emit the 16 block lists (order preserving) ----
  const int ttx = tile % gx, tty = tile / gx;
  const float tx0 = (float)(ttx * TILE), ty0 = (float)(tty * TILE);
  if (tid < NLIST) run[tid] = 0;
  __syncthreads();
  uint2* sub = b.sublist + (size_t)NLIST * start;
  for (int base = 0; base < len; base += 256) {
    const int i = base + tid;
    const bool have = i < len;
    uint32_t id = 0, pidx = 0, mask = 0, rec0 = 0;
    BlkRect br = {0, 0, 0, 0};
    if (have) {
      id = (uint32_t)(in_lds ? sk[i] : gk[i]);
      const float4* sp = (const float4*)(g.splat + (size_t)id * SPLAT_F);
      const float4 A = sp[0], B = sp[1];
      // alpha >= 1/255  <=>  d^T Q d <= 2 tau, tau = ln(255 o), Q = [[A.z, A.w],[A.w, B.x]]
      const float tau = __logf(255.f * B.y);
      const float det = A.z * B.x - A.w * A.w;
      if (det > 0.f) {
        // a 4x4 block (pixel centres [x0, x0+3] x [y0, y0+3]) is listed only where the {alpha >= 1/255} region can reach:
        // its axis-aligned bound must overlap the block AND the block must come within sqrt(2 tau lambda_max) of the centre
        // (exact for isotropic splats, where the box test alone keeps the corners a disc cannot reach).  Both necessary.
        const float t2 = 2.f * fmaxf(tau, 0.f);
        const float k = t2 / det;
        const float hx = sqrtf(k * B.x) * 1.0002f + 0.002f;
        const float hy = sqrtf(k * A.z) * 1.0002f + 0.002f;
        const float sxx = B.x / det, syy = A.z / det, mid = 0.5f * (sxx + syy);
        const float lam = mid + sqrtf(fmaxf(mid * mid - 1.f / det, 0.f));
        const float r2 = t2 * lam * 1.0004f + 0.01f;
        const float cx = A.x - tx0, cy = A.y - ty0;
        const bool live = tau > 0.f;
        bool bx[4], by[4];
        float ex[4], ey[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float lo = 4.f * q, hi = 4.f * q + 3.f;
          bx[q] = live && (cx - hx <= hi) && (cx + hx >= lo);
          by[q] = live && (cy - hy <= hi) && (cy + hy >= lo);
          const float dxq = fmaxf(fmaxf(lo - cx, cx - hi), 0.f), dyq = fmaxf(fmaxf(lo - cy, cy - hi), 0.f);
          ex[q] = dxq * dxq; ey[q] = dyq * dyq;
        }
#pragma unroll
        for (int my = 0; my < 4; my++)
#pragma unroll
          for (int kx = 0; kx < 4; kx++) {
            const int L = 4 * ((my >> 1) * 2 + (kx >> 1)) + (my & 1) * 2 + (kx & 1);   // 4 * sub-tile + block in sub-tile
            if (bx[kx] && by[my] && (ex[kx] + ey[my] <= r2)) mask |= 1u << L;
          }
      } else {
        mask = 0xffffu;  // degenerate conic: no culling, the exact per-pixel rule decides
      }
      // pair index of (Gaussian, tile) in Gaussian-major order (-> submask), and the splat's first gradient record
      const uint32_t r0 = g.rect[(size_t)id * 2], r1 = g.rect[(size_t)id * 2 + 1];
      const int minx = r0 & 0xffff, miny = r0 >> 16, rw = (int)(r1 & 0xffff) - minx;
      pidx = g.block_tiles[id >> 8] + g.tileoff[id] + (uint32_t)((tty - miny) * rw + (ttx - minx));
      br = block_rect(A, B, r0, r1);
      rec0 = g.block_blk[id >> 8] + g.blkoff[id];
      // blocks outside the block rectangle cannot be listed (it bounds the same region with slack); belt and braces
#pragma unroll
      for (int L = 0; L < NLIST; L++) {
        const int bx = ttx * 4 + ((L >> 2) & 1) * 2 + (L & 1) - br.bx0, by = tty * 4 + (L >> 3) * 2 + ((L >> 1) & 1) - br.by0;
        if (bx < 0 || by < 0 || bx >= br.bw || by >= br.bh) mask &= ~(1u << L);
      }
      if (pidx < N_cap && (size_t)rec0 + (size_t)br.bw * br.bh <= (size_t)NLIST * N_cap) b.submask[pidx] = (uint16_t)mask;
      else mask = 0;   // only on capacity overflow (flagged in the header)
    }
    unsigned long long bal[NLIST];
#pragma unroll
    for (int L = 0; L < NLIST; L++) bal[L] = __ballot((mask >> L) & 1u);
    if (lane < NLIST) {
      unsigned long long mine = bal[0];
#pragma unroll
      for (int L = 1; L < NLIST; L++) mine = lane == L ? bal[L] : mine;
      wcnt[wv][lane] = __popcll(mine);
    }
    __syncthreads();
    if (tid < 4 * NLIST) {
      const int w = tid >> 4, L = tid & 15;
      uint32_t p0 = run[L];
      for (int w2 = 0; w2 < w; w2++) p0 += wcnt[w2][L];
      pre[w][L] = p0;
    }
    __syncthreads();
    if (mask) {
      const unsigned long long lt = (1ull << lane) - 1ull;
      // entry = {splat id, gradient record of (splat, block)}: row-major position of the block in the splat's rectangle
      const uint32_t recT = rec0 + (uint32_t)((tty * 4 - br.by0) * br.bw + (ttx * 4 - br.bx0));
#pragma unroll
      for (int L = 0; L < NLIST; L++)
        if ((mask >> L) & 1u) {
          const uint32_t rec = recT + (uint32_t)(((L >> 3) * 2 + ((L >> 1) & 1)) * br.bw + ((L >> 2) & 1) * 2 + (L & 1));
          sub[(size_t)L * len + pre[wv][L] + __popcll(bal[L] & lt)] = make_uint2(id, rec);
        }
    }
    __syncthreads();
    if (tid < NLIST) run[tid] = pre[3][tid] + wcnt[3][tid];
    __syncthreads();
  }
  if (tid < NLIST) iv.subcount[NLIST * tile + tid] = run[tid];
  return true;
}

