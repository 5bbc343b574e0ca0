// Tile binning for gfx950, designed around the 160 KB LDS instead of a global multi-pass radix sort:
//   1. preprocess counted overlaps per tile (tile_count[T], fire-and-forget atomics);
//   2. scan_tiles: one workgroup turns tile_count into ranges[T+1] (num_rendered = ranges[T]) and seeds cursor[T];
//   3. scatter: every visible Gaussian drops a 64-bit key (depth_bits << 32 | id) into each overlapped tile's bin
//      (slot = returning atomic on the tile cursor) -- order inside a bin is arbitrary at this point;
//   4. sort_tiles: one workgroup per tile sorts its bin in LDS (bitonic, keys unique => deterministic) and writes
//      the depth-ordered id list.  Ordering == (depth, Gaussian index), the order a stable sort on the lineage's
//      (tile | depth) keys produces.
// HBM traffic is 12 B/pair written + 12 B/pair read + 4 B/pair written, versus 24 B/pair x 6 radix passes.
#include "mm3dgs_common.h"

// ---- 2. scan --------------------------------------------------------------------------------------------------
#define SCAN_BLOCK 1024
__global__ void __launch_bounds__(SCAN_BLOCK) scan_tiles_kernel(int T, ImageView iv) {
  __shared__ uint32_t wave_tot[SCAN_BLOCK / 64];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t maxlen_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { carry_s = 0; maxlen_s = 0; }
  __syncthreads();
  uint32_t local_max = 0;
  for (int base = 0; base < T; base += SCAN_BLOCK) {
    int i = base + tid;
    uint32_t v = (i < T) ? iv.tile_count[i] : 0u;
    local_max = max(local_max, v);
    // inclusive scan inside the wave (Hillis-Steele over lanes)
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t prefix = carry_s;
    for (int w = 0; w < wv; w++) prefix += wave_tot[w];
    uint32_t excl = prefix + x - v;
    if (i < T) { iv.ranges[i] = excl; iv.cursor[i] = excl; }
    __syncthreads();
    if (tid == SCAN_BLOCK - 1) carry_s = prefix + x;
    __syncthreads();
  }
  atomicMax(&maxlen_s, local_max);
  __syncthreads();
  if (tid == 0) {
    iv.ranges[T] = carry_s;
    iv.hdr->num_rendered = carry_s;
    iv.hdr->max_tile_len = maxlen_s;
  }
}
void launch_scan_tiles(int T, ImageView iv, hipStream_t s) {
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, T, iv);
}

// ---- 3. scatter -----------------------------------------------------------------------------------------------
// One lane per Gaussian, two sweeps over its tile rectangle around a per-workgroup LDS histogram:
//   sweep 1 counts the workgroup's overlaps per tile in LDS; one returning global atomic per *touched tile* reserves
//   a contiguous span in that tile's bin; sweep 2 hands out slots inside the spans with returning LDS atomics.
// Spatially coherent Gaussian order (SLAM maps) turns ~2.3 global atomics per Gaussian into a few per workgroup.
// A Gaussian whose rectangle covers more than 32 tiles is spread over the whole wave (rectangle broadcast with
// readlane) so that one huge splat does not serialise a wave for thousands of iterations.
template <bool WRITE>
__device__ __forceinline__ void sweep_rect(uint32_t* cnt, int gx, int minx, int miny, int w, int area, unsigned long long key,
                                           unsigned long long* keys, uint32_t N_cap, int lane) {
  unsigned long long big = __ballot(area > 32);
  if (area > 0 && area <= 32) {
    for (int k = 0; k < area; k++) {
      uint32_t slot = atomicAdd(&cnt[(miny + k / w) * gx + minx + k % w], 1u);
      if (WRITE && slot < N_cap) keys[slot] = key;
    }
  }
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const int sminx = __builtin_amdgcn_readlane(minx, src), sminy = __builtin_amdgcn_readlane(miny, src);
    const int sw = __builtin_amdgcn_readlane(w, src), sarea = __builtin_amdgcn_readlane(area, src);
    const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)key, src), khi = __builtin_amdgcn_readlane((uint32_t)(key >> 32), src);
    const unsigned long long skey = ((unsigned long long)khi << 32) | klo;
    for (int k = lane; k < sarea; k += 64) {
      uint32_t slot = atomicAdd(&cnt[(sminy + k / sw) * gx + sminx + k % sw], 1u);
      if (WRITE && slot < N_cap) keys[slot] = skey;
    }
  }
}

__global__ void __launch_bounds__(256)
scatter_keys_kernel(int P, int gx, int T, GeomView g, ImageView iv, BinView b, uint32_t N_cap, int lds_tiles) {
  extern __shared__ uint32_t hist[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int t = tid; t < lds_tiles; t += 256) hist[t] = 0;
  int idx = blockIdx.x * 256 + tid;
  uint32_t r0 = 0, r1 = 0, dbits = 0;
  if (idx < P) {
    r0 = g.rect[(size_t)idx * 2];
    r1 = g.rect[(size_t)idx * 2 + 1];
    if (r1 != r0) dbits = __float_as_uint(g.depth[idx]);
  }
  const int minx = r0 & 0xffff, miny = r0 >> 16, maxx = r1 & 0xffff, maxy = r1 >> 16;
  const int w = maxx - minx, h = maxy - miny;
  const int area = (w > 0 && h > 0) ? w * h : 0;
  const unsigned long long key = ((unsigned long long)dbits << 32) | (uint32_t)idx;
  if (lds_tiles) {
    __syncthreads();
    sweep_rect<false>(hist, gx, minx, miny, w, area, key, b.keys, N_cap, lane);
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
      uint32_t c = hist[t];
      if (c) hist[t] = atomicAdd(&iv.cursor[t], c);
    }
    __syncthreads();
    sweep_rect<true>(hist, gx, minx, miny, w, area, key, b.keys, N_cap, lane);
  } else {
    sweep_rect<true>(iv.cursor, gx, minx, miny, w, area, key, b.keys, N_cap, lane);
  }
  if (idx == 0 && iv.hdr->num_rendered > N_cap) iv.hdr->overflow = 1;
}

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

// Tier kernel: handles tiles with lo < len <= CAP in LDS; when GLOBAL_TAIL it also sorts len > CAP in place in
// global memory (rare: > 16 K splats on one tile).
template <int CAP, bool GLOBAL_TAIL>
__global__ void __launch_bounds__(256) sort_tiles_kernel(int T, int lo, ImageView iv, BinView b, uint32_t N_cap) {
  __shared__ unsigned long long sk[CAP];
  // XCD-aware tile order is not needed here: a tile's bin is private to its workgroup.
  int tile = blockIdx.x;
  if (tile >= T) return;
  uint32_t start = min(iv.ranges[tile], N_cap), end = min(iv.ranges[tile + 1], N_cap);
  int len = (int)(end - start);
  if (len <= lo) return;
  const int tid = threadIdx.x;
  unsigned long long* gk = b.keys + start;
  uint32_t* pl = b.point_list + start;
  if (len <= CAP) {
    for (int i = tid; i < len; i += 256) sk[i] = gk[i];
    __syncthreads();
    if (len > 1) bitonic_any_len([&](int i) -> unsigned long long& { return sk[i]; }, len, tid, 256);
    for (int i = tid; i < len; i += 256) pl[i] = (uint32_t)sk[i];
  } else if (GLOBAL_TAIL) {
    __syncthreads();
    bitonic_any_len([&](int i) -> unsigned long long& { return gk[i]; }, len, tid, 256);
    for (int i = tid; i < len; i += 256) pl[i] = (uint32_t)gk[i];
  }
}

#define SORT_CAP_SMALL 2048   // 16 KB LDS: the common case (SLAM lists are a few hundred entries)
#define SORT_CAP_LARGE 16384  // 128 KB LDS: one workgroup per CU

void launch_scatter_sort(const CamDev& cam, int P, GeomView g, ImageView iv, BinView b, size_t N_cap,
                         const int32_t*, hipStream_t s) {
  int T = cam.gx * cam.gy;
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  const int lds_tiles = T <= MAX_LDS_TILES ? T : 0;
  if (P > 0)
    hipLaunchKernelGGL(scatter_keys_kernel, dim3((P + 255) / 256), dim3(256), (size_t)lds_tiles * 4, s, P, cam.gx, T, g, iv, b,
                       ncap, lds_tiles);
  hipLaunchKernelGGL((sort_tiles_kernel<SORT_CAP_SMALL, false>), dim3(T), dim3(256), 0, s, T, 0, iv, b, ncap);
  hipLaunchKernelGGL((sort_tiles_kernel<SORT_CAP_LARGE, true>), dim3(T), dim3(256), 0, s, T, SORT_CAP_SMALL, iv, b, ncap);
}
