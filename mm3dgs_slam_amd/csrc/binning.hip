// Tile binning for gfx950, designed around the 160 KB LDS instead of a global multi-pass radix sort:
//   1. preprocess counted overlaps per tile (tile_count[T], fire-and-forget atomics);
//   2. scan_tiles: one workgroup turns tile_count into ranges[T+1] (num_rendered = ranges[T]) and seeds cursor[T];
//   3. scatter: every visible Gaussian drops a 64-bit key (depth_bits << 32 | id) into each overlapped tile's bin
//      (slot = returning atomic on the tile cursor) -- order inside a bin is arbitrary at this point;
//   4. sort_tiles (sort_tile.h): one workgroup per tile sorts its bin in LDS (register bitonic runs + rank merge, keys
//      unique => deterministic).  Ordering == (depth, Gaussian index), the order a stable sort on the lineage's
//      (tile | depth) keys produces.  The same workgroup then splits the tile's list into sixteen depth-ordered lists, one
//      per 4x4-pixel block: a splat is listed for a block only if the bound of { alpha >= 1/255 } reaches it (conservative,
//      so compositing the block lists equals compositing the full tile list); the bin is left in sorted order as
//      (block mask | per-tile gradient record << 32) per pair -- what the backward compositor's per-tile combine needs to find
//      the block records of a pair by list position.  In the SLAM path steps 2-3 are one launch (scatter_scan_kernel) and step 4 runs inside the
//      forward compositing launch (composite.hip).
// HBM traffic is ~60 B/pair touched, versus 24 B/pair x 6 passes for a global radix sort of 64-bit keys.
#include "mm3dgs_common.h"
#include "sort_tile.h"

// ---- 2. scan --------------------------------------------------------------------------------------------------
#define SCAN_BLOCK 1024
// exclusive scan of src[0..n) by one 1024-lane workgroup; returns the total (valid in every lane)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t* src, uint32_t* dst0, uint32_t* dst1, int n, uint32_t* wave_tot,
                                                    uint32_t* carry_s, uint32_t* maxv, bool clear_src) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) *carry_s = 0;
  __syncthreads();
  uint32_t local_max = 0;
  for (int base = 0; base < n; base += SCAN_BLOCK) {
    int i = base + tid;
    uint32_t v = (i < n) ? src[i] : 0u;
    if (clear_src && i < n) src[i] = 0u;   // leave the counters zero for the next forward (persistent state buffers)
    local_max = max(local_max, v);
    const uint32_t x = wave_scan_incl(v);  // inclusive scan inside the wave
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t prefix = *carry_s;
    for (int w = 0; w < wv; w++) prefix += wave_tot[w];
    uint32_t excl = prefix + x - v;
    if (i < n) { dst0[i] = excl; if (dst1) dst1[i] = excl; }
    __syncthreads();
    if (tid == SCAN_BLOCK - 1) *carry_s = prefix + x;
    __syncthreads();
  }
  if (maxv) atomicMax(maxv, local_max);
  __syncthreads();
  return *carry_s;
}

// sticky: persistent state buffers (fused SLAM path) -- overflow / max_tile_len / max_num_rendered accumulate over forwards
// until the host clears them, so one header read after an optimisation loop sees an overflow of any of its iterations
__global__ void __launch_bounds__(SCAN_BLOCK) scan_tiles_kernel(int T, int nblocks, GeomView g, ImageView iv, int sticky) {
  __shared__ uint32_t wave_tot[SCAN_BLOCK / 64];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t maxlen_s;
  if (threadIdx.x == 0) maxlen_s = 0;
  uint32_t total = block_excl_scan(iv.tile_count, iv.ranges, iv.cursor, T, wave_tot, &carry_s, &maxlen_s, true);
  if (threadIdx.x == 0) {
    iv.ranges[T] = total;
    iv.hdr->num_rendered = total;
    iv.hdr->max_tile_len = sticky ? max(iv.hdr->max_tile_len, maxlen_s) : maxlen_s;
    iv.hdr->max_num_rendered = sticky ? max(iv.hdr->max_num_rendered, total) : total;
    if (!sticky) iv.hdr->overflow = 0;
    iv.hdr->bin_cap = 0;
  }
  __syncthreads();
  // tiles touched per preprocess workgroup -> exclusive prefix (first pair index of each workgroup's Gaussians)
  uint32_t tot2 = block_excl_scan(g.block_tiles, g.block_tiles, nullptr, nblocks, wave_tot, &carry_s, nullptr, false);
  if (threadIdx.x == 0) g.block_tiles[nblocks] = tot2;
}
void launch_scan_tiles(int T, int P, GeomView g, ImageView iv, hipStream_t s, int sticky) {
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, T, (P + 255) / 256, g, iv, sticky);
}

// ---- 3. scatter -----------------------------------------------------------------------------------------------
// One lane per Gaussian, two sweeps over its tile rectangle around a per-workgroup LDS histogram:
//   sweep 1 counts the workgroup's overlaps per tile in LDS; one returning global atomic per *touched tile* reserves
//   a contiguous span in that tile's bin; sweep 2 hands out slots inside the spans with returning LDS atomics.
// Spatially coherent Gaussian order (SLAM maps) turns ~2.3 global atomics per Gaussian into a few per workgroup.
// A Gaussian whose rectangle covers more than 32 tiles is spread over the whole wave (rectangle broadcast with
// readlane) so that one huge splat does not serialise a wave for thousands of iterations.
// (Measured and rejected in round 6: the pair's Gaussian-major index in the key's low word and payload[pair] = id | block mask written HERE, so that the
//  sort's emission reads one word per sorted entry instead of gathering five arrays by id: the per-pair mask arithmetic runs at a few lanes per wave in
//  this sweep -- scatter 105 -> 209 us at 1080p / 3 M Gaussians for 406 -> 340 us of sort.)
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

// ---- 3b. scatter with the scan folded in (persistent-state SLAM path) ----------------------------------------------
// Every workgroup scans the T tile counters itself (4.8 KB from L2 at 640x480, a few microseconds) instead of waiting for a
// one-workgroup scan kernel: one launch and one dependent-launch gap less per render.  Workgroup 0 publishes ranges[] and
// the header for the later kernels, workgroup 1 turns the per-preprocess-workgroup tile totals into their exclusive prefix
// (consumed by sort_tiles).  cursor[] counts from zero here; sort_tiles leaves cursor[] and tile_count[] zero again.
__device__ __forceinline__ uint32_t block256_excl_scan_inplace(uint32_t* a, int n, uint32_t* wave_tot, uint32_t* maxv) {
  // a[0..n) in LDS or global, 256 lanes, each lane owns a contiguous span; returns the total; a <- exclusive prefix
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + 255) / 256;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  uint32_t sum = 0, mx = 0;
  for (int i = lo; i < hi; i++) { const uint32_t v = a[i]; sum += v; mx = max(mx, v); }
  const uint32_t x = wave_scan_incl(sum);
  if (lane == 63) wave_tot[wv] = x;
  if (maxv) atomicMax(maxv, mx);
  __syncthreads();
  uint32_t pre = x - sum;
  for (int w = 0; w < wv; w++) pre += wave_tot[w];
  const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  for (int i = lo; i < hi; i++) { const uint32_t v = a[i]; a[i] = pre; pre += v; }
  __syncthreads();
  return total;
}

__global__ void __launch_bounds__(256)
scatter_scan_kernel(int P, int gx, int T, int nblocks_pre, GeomView g, ImageView iv, BinView b, uint32_t N_cap) {
  extern __shared__ uint32_t sh[];
  uint32_t* hist = sh;          // [T]   per-workgroup overlap counts, then span bases
  uint32_t* rng = sh + T;       // [T]   exclusive scan of the global tile counters
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t maxlen_s;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) maxlen_s = 0;
  // this lane's Gaussian: loads issued first, they land while the tile counters are scanned
  const int idx = blockIdx.x * 256 + tid;
  uint32_t r0 = 0, r1 = 0, dbits = 0;
  if (idx < P) {
    r0 = g.rect[(size_t)idx * 2];
    r1 = g.rect[(size_t)idx * 2 + 1];
    dbits = __float_as_uint(g.depth[idx]);
  }
  for (int t = tid; t < T; t += 256) { hist[t] = 0; rng[t] = iv.tile_count[t]; }
  __syncthreads();
  const uint32_t total = block256_excl_scan_inplace(rng, T, wave_tot, &maxlen_s);
  if (blockIdx.x == 0) {
    for (int t = tid; t < T; t += 256) iv.ranges[t] = rng[t];
    if (tid == 0) {
      iv.ranges[T] = total;
      iv.hdr->num_rendered = total;
      // sticky (this kernel only runs on persistent state): the host clears these three after it has read them
      iv.hdr->max_tile_len = max(iv.hdr->max_tile_len, maxlen_s);
      iv.hdr->max_num_rendered = max(iv.hdr->max_num_rendered, total);
      if (total > N_cap) iv.hdr->overflow = 1u;
      iv.hdr->bin_cap = 0;   // packed bins
    }
  }
  if (blockIdx.x == (gridDim.x > 1 ? 1 : 0)) {
    __syncthreads();
    const uint32_t tot2 = block256_excl_scan_inplace(g.block_tiles, nblocks_pre, wave_tot, nullptr);
    if (tid == 0) g.block_tiles[nblocks_pre] = tot2;
  }
  if (r1 == r0) dbits = 0;   // culled: depth[] was not written this frame
  const int minx = r0 & 0xffff, miny = r0 >> 16, maxx = r1 & 0xffff, maxy = r1 >> 16;
  const int w = maxx - minx, h = maxy - miny;
  const int area = (w > 0 && h > 0) ? w * h : 0;
  const unsigned long long key = ((unsigned long long)dbits << 32) | (uint32_t)idx;
  sweep_rect<false>(hist, gx, minx, miny, w, area, key, b.keys, N_cap, lane);
  __syncthreads();
  for (int t = tid; t < T; t += 256) {
    uint32_t c = hist[t];
    if (c) hist[t] = rng[t] + atomicAdd(&iv.cursor[t], c);
  }
  __syncthreads();
  sweep_rect<true>(hist, gx, minx, miny, w, area, key, b.keys, N_cap, lane);
}

// ---- 4. per-tile sort (body in sort_tile.h) --------------------------------------------------------------------------
template <int CAP, bool GLOBAL_TAIL>
__global__ void __launch_bounds__(256)
sort_tiles_kernel(int T, int gx, int lo, GeomView g, ImageView iv, BinView b, uint32_t N_cap, int clean, int ex) {      // (ex: the probe word of -DMM3DGS_PROBES builds, else 0)
  __shared__ unsigned long long sk[CAP];
  __shared__ SortShared sh;
  __shared__ SortEmit em;
  const int tile = blockIdx.x;
  if (tile >= T) return;
  sort_tile_body<CAP, GLOBAL_TAIL>(tile, gx, lo, g, iv, b, N_cap, clean, sk, sh, ex, 0, 0, DIRECT_SLOT_BITS_MAX, &em);
}

#define SORT_CAP_SMALL 2048   // 16 KB LDS: the common case (SLAM lists are a few hundred entries)
#define SORT_CAP_LARGE 16384  // 128 KB LDS: one workgroup per CU

void launch_scatter_sort(const CamDev& cam, int P, GeomView g, ImageView iv, BinView b, size_t N_cap,
                         const int32_t*, hipStream_t s, bool scatter_only) {
  int T = cam.gx * cam.gy;
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  const int lds_tiles = T <= MAX_LDS_TILES ? T : 0;
  // persistent state: the sort leaves cursor[] (and tile_count[]) zero -- the direct-bin path of the NEXT forward counts from zero.
  // (Without the fused scan, scan_tiles seeds cursor[] with the range starts and the scatter leaves the range ends there: on tile grids
  //  above MAX_FUSED_SCAN_TILES a following direct-bin forward read those as pair counts -- garbage ids, out-of-bounds records.)
  const int clean = (cam.fused_scan || cam.state_clean) ? 1 : 0;
  if (cam.fused_scan)   // P > 0 and T <= MAX_FUSED_SCAN_TILES guaranteed by the caller
    hipLaunchKernelGGL(scatter_scan_kernel, dim3((P + 255) / 256), dim3(256), (size_t)T * 8, s, P, cam.gx, T, (P + 255) / 256, g, iv, b, ncap);
  else if (P > 0)
    hipLaunchKernelGGL(scatter_keys_kernel, dim3((P + 255) / 256), dim3(256), (size_t)lds_tiles * 4, s, P, cam.gx, T, g, iv, b,
                       ncap, lds_tiles);
  if (scatter_only) return;   // the caller sorts inside its compositing launch
  if (cam.sort_single) {
    hipLaunchKernelGGL((sort_tiles_kernel<SORT_CAP_SMALL, true>), dim3(T), dim3(256), 0, s, T, cam.gx, 0, g, iv, b, ncap, clean, (int)PROBE_WORD(cam));
  } else {
    hipLaunchKernelGGL((sort_tiles_kernel<SORT_CAP_SMALL, false>), dim3(T), dim3(256), 0, s, T, cam.gx, 0, g, iv, b, ncap, clean, (int)PROBE_WORD(cam));
    hipLaunchKernelGGL((sort_tiles_kernel<SORT_CAP_LARGE, true>), dim3(T), dim3(256), 0, s, T, cam.gx, SORT_CAP_SMALL, g, iv, b, ncap, 0, (int)PROBE_WORD(cam));
  }
}

// ---- load-balanced workgroup -> tile table for the SLAM compositors ----------------------------------------------------------------
// The compositors launch one 256-lane workgroup per tile, and a 640x480 grid (1200 tiles) fits the chip in ONE round: measured placement
// (tools/ubench/placement.hip): workgroup b runs on XCD b % 8; inside an XCD the k-th of its workgroups goes to CU slot k % 32, so with
// 150 workgroups per XCD 22 CUs hold five tiles and 10 hold four -- and the launch ends when the CUs with five are done (per-CU work
// max / mean 1.19 on the benchmark scene, tools/list_balance.py).  This kernel keeps every XCD's contiguous span of tiles (its L2
// locality) but deals the tiles of the span to the XCD's workgroup slots by load: the heaviest tiles to the CU slots that hold one tile
// less, the rest serpentine over the others (max / mean ~1.04).  Load = the wave steps of the tile's last render (a wave's four rows
// advance together: sum over the four 8x8 sub-tiles of the longest of their four block lists) -- from whatever view was rendered last;
// the table is a permutation whatever the loads are, so a stale or meaningless load only costs speed.  One workgroup per XCD.
#define ORDER_MAX_PER 160     // five rounds of 32 CU slots: beyond that the workgroups of a launch are placed dynamically
// Round 4 experiment, measured and NOT adopted (its branch left the sources in round 5): the eight spans cut by LOAD instead of
// by tile count.  On a camera that moves, keyframes seed the newly seen side of the image and the tile lists there grow: with equal-count
// spans one XCD carries 1.05 - 1.11 x the mean wave steps (tools/xcd_balance.py on the bench's `moving` scenario; 1.02 - 1.03 on the bounded
// trajectory), load-cut contiguous spans bring that to 1.005 (a span then holds up to TILE_SPAN_SLOTS tiles, the compositors launch
// 8 x TILE_SPAN_SLOTS workgroups and the ones beyond an XCD's span leave at once).  On the MI355X the two builds are indistinguishable on the
// bounded trajectory (31.33 vs 31.32 frames/s under rocprofv3) and the load-cut one is 1.2 % SLOWER on the moving one (42.16 vs 41.67 ms per
// frame): as with round 3's variants of the dealing, a launch's duration does not follow the per-XCD / per-CU sum of wave steps closely
// enough for a better model balance to show.  The default stays: equal-count spans, the launch's grid = the tiles.
__global__ void __launch_bounds__(256) tile_order_kernel(ImageView iv, int T, uint32_t key) {
  __shared__ uint32_t pre[8 * ORDER_MAX_PER + 1];        // inclusive prefix of the loads in tile order, pre[-1] = 0 at pre[0]
  __shared__ uint32_t load[TILE_SPAN_SLOTS];
  __shared__ uint32_t wsum[4];
  __shared__ int cut[9];
  const int x = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // loads of all tiles: thread t owns the CH consecutive tiles [t CH, (t + 1) CH)
  const int CH = (T + 255) >> 8;
  uint32_t mine_sum = 0;
  uint32_t l5[(8 * ORDER_MAX_PER + 255) / 256];
#pragma unroll
  for (int u = 0; u < (8 * ORDER_MAX_PER + 255) / 256; u++) {
    const int tile = tid * CH + u;
    uint32_t m = 0;
    if (u < CH && tile < T) {
      const uint4* sc = (const uint4*)(iv.subcount + (size_t)tile * NLIST);
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint4 q = sc[w];
        m += max(max(q.x, q.y), max(q.z, q.w));        // load = wave steps (row steps as the load measured the same: DESIGN.md section 4)
      }
      m += 1u;                                         // (a real tile outranks the padding; an unrendered grid cuts into equal spans)
    }
    l5[u] = m;
    mine_sum += m;
  }
  // exclusive scan of the per-thread sums over the workgroup
  const uint32_t incl = wave_scan_incl(mine_sum);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t base = incl - mine_sum;
  for (int q = 0; q < wv; q++) base += wsum[q];
  if (tid == 0) pre[0] = 0u;
  {
    uint32_t run = base;
#pragma unroll
    for (int u = 0; u < (8 * ORDER_MAX_PER + 255) / 256; u++) {
      const int tile = tid * CH + u;
      run += l5[u];
      if (u < CH && tile < T) pre[tile + 1] = run;
    }
  }
  __syncthreads();
  // cut k: equal-count spans, the arithmetic map's (spans of equal LOAD measured no better on a bounded trajectory and 1.2 % slower on a
  // moving one: DESIGN.md section 4); every workgroup computes all nine identically
  if (tid < 9) cut[tid] = tid == 8 ? T : min(T, tid * ((T + 7) >> 3));
  __syncthreads();
  // a span that would not fit its XCD's workgroup slots: fall back to equal-count spans (the same decision in every workgroup)
  bool fits = true;
  for (int k = 0; k < 8; k++) fits = fits && (cut[k + 1] - cut[k] <= TILE_SPAN_SLOTS) && (cut[k + 1] >= cut[k]);
  const int per_eq = (T + 7) >> 3;
  const int lo_t = fits ? cut[x] : min(T, x * per_eq), hi_t = fits ? cut[x + 1] : min(T, (x + 1) * per_eq);
  const int per = hi_t - lo_t;                        // tiles of this XCD's span
  const int j = tid;                                  // tile lo_t + j of the span
  const uint32_t mine = j < per ? pre[lo_t + j + 1] - pre[lo_t + j] : 0u;
  load[j] = mine;
  __syncthreads();
  // the slots of this XCD beyond its span: no tile
  for (int i = per + tid; i < slam_span_slots(T); i += 256) iv.tile_order[(size_t)i * 8 + x] = 0u;
  if (x == 0 && tid == 0) {
    iv.hdr->tile_order_tiles = key;      // (the image size, not T: the table's offset in image_state depends on H * W)
    iv.hdr->mean_wave_steps = T > 0 ? (pre[T] - (uint32_t)T) / (4u * (uint32_t)T) : 0u;      // (loads carry + 1 per tile; four waves per tile)
  }
  if (j >= per) return;
  int rank = 0;                                        // descending load, ties by index: a permutation of [0, per)
  for (int k = 0; k < per; k++) {
    const uint32_t o = load[k];
    rank += (o > mine || (o == mine && k < j)) ? 1 : 0;
  }
  const int R = (per + 31) >> 5;                       // rounds; CU slots c < full hold R workgroups, the others R - 1
  const int full = per - (R - 1) * 32, L = 32 - full, n_light = L * (R - 1);
  int r, c;
  if (rank < n_light) {                                // heaviest tiles: the slots with one workgroup less, serpentine over their R - 1 rounds
    r = rank / L;
    const int pos = rank - r * L;
    c = full + ((r & 1) ? L - 1 - pos : pos);
  } else {
    const int q = rank - n_light;
    r = q / full;
    const int pos = q - r * full;
    c = (r & 1) ? full - 1 - pos : pos;
  }
  const int i = r * 32 + c;                            // index of the workgroup inside the XCD: blockIdx = 8 i + x
  iv.tile_order[(size_t)i * 8 + x] = (uint32_t)(lo_t + j) + 1u;
}
// Grids of several rounds of workgroups (configs[3]: 1200x680 = 3225 tiles = 2.5 rounds at five workgroups per CU; up to 11264 tiles): which CU slot a
// workgroup lands on is decided by the dispatcher as slots free up, so there is nothing to deal, and the arithmetic map stays (round 5 measured the one order a
// table could still impose -- every XCD's span by descending load, longest-processing-time-first -- at 795 k Gaussians: 7.98 frames/s against 8.08, the
// compositors 1.5 % slower: neighbouring tiles share splats, and the order that balances the tail scatters them over time; profiles/r05_c4_lpt_order.txt).
// What this kernel leaves is the mean list walk the compositing waves' priorities refer to.
__global__ void __launch_bounds__(256) mean_wave_steps_kernel(ImageView iv, int T) {
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x;
  uint32_t sum = 0;
  for (int t = tid; t < T; t += 256) {
    const uint4* sc = (const uint4*)(iv.subcount + (size_t)t * NLIST);
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint4 q = sc[w]; sum += max(max(q.x, q.y), max(q.z, q.w)); }
  }
  sum = wave_scan_incl(sum);
  if ((tid & 63) == 63) wsum[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) iv.hdr->mean_wave_steps = T > 0 ? (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (4u * (uint32_t)T) : 0u;
}
bool launch_tile_order(int T, int H, int W, ImageView iv, hipStream_t s) {
  const int per = (T + 7) >> 3;
  if (T < 64 || tile_order_key(H, W) == 0u) return false;     // (tiny grids: nothing to balance)
  if (per > ORDER_MAX_PER) {                                   // (several rounds of dynamic placement: no table, only the priorities' reference)
    hipLaunchKernelGGL(mean_wave_steps_kernel, dim3(1), dim3(256), 0, s, iv, T);
    return false;
  }
  hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(256), 0, s, iv, T, tile_order_key(H, W));
  return true;
}
