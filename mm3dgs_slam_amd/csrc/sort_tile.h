// Per-tile depth sort + emission of the sixteen 4x4-block lists, as a device function shared by sort_tiles_kernel
// (binning.hip) and the fused sort + forward-composite kernel (composite.hip).
#pragma once
#include "mm3dgs_common.h"
#include "tile_mask.h"

struct SortEmit {              // scratch of the barrier-synchronised emission (packed bins, lists beyond the rank sort): only alive inside sort_tile_body
  uint32_t wcnt[4][NLIST];     // per-wave entry counts of a 256-entry chunk, per block list
  uint32_t pre[4][NLIST];      // write cursor of (wave, list) for the chunk
};
struct SortShared {            // LDS of one sorting workgroup besides the key array
  uint32_t run[NLIST];         // entries emitted so far per list (the final list lengths)
  uint32_t start, len;         // the tile's bin (for the compositing phase of the same workgroup)
  uint32_t scan_tot[4];        // direct bins: wave totals of the pair count (tile 0's workgroup)
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

// ---- in-register bitonic sort of one 64-key run per wave ------------------------------------------------------------
// The partner of a compare-exchange at distance J comes through the VALU's own lane crossbars instead of the LDS one (ds_bpermute_b32:
// 26 ns of latency per dependent step, 10 ns of LDS pipe per instruction at 4 waves per SIMD; v_mov_b32_dpp: 2 ns, tools/ubench): distances
// 1, 2 (quad_perm) and 8 (row_ror:8) are one DPP move per dword, 4 is two (mirror of the 8-lane half, then of the quads), 16 and 32 are
// gfx950's v_permlane16_swap / v_permlane32_swap of two copies of the dword (before round 4: ds_bpermute exchanges, sort + forward +0.9 us).  Any correct network yields the same order (keys are unique).
template <int J>
__device__ __forceinline__ uint32_t lane_xor_dpp(uint32_t v) {
  static_assert(J == 1 || J == 2 || J == 4 || J == 8, "distances inside a 16-lane row");
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);         // quad_perm:[1,0,3,2]
  else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm:[2,3,0,1]
  else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);   // row_ror:8
  else {
    const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);                              // row_half_mirror: i -> 7 - i
    return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, true);                                 // quad_perm:[3,2,1,0]: -> i ^ 4
  }
}
template <int J>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long key, int lane) {
  uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  if constexpr (J <= 8) {
    lo = lane_xor_dpp<J>(lo); hi = lane_xor_dpp<J>(hi);
  } else if constexpr (J == 16) {
    // [0]: the even rows' values in both rows of a pair, [1]: the odd rows' (rows = 16 lanes)
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    lo = (lane & 16) ? a[0] : a[1]; hi = (lane & 16) ? b[0] : b[1];
  } else {
    static_assert(J == 32, "wave64");
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    lo = (lane & 32) ? a[0] : a[1]; hi = (lane & 32) ? b[0] : b[1];
  }
  return ((unsigned long long)hi << 32) | lo;
}
template <int K, int J>
__device__ __forceinline__ unsigned long long bitonic_step64(unsigned long long key, int lane) {
  const unsigned long long other = lane_xor64<J>(key, lane);
  const bool up = (lane & K) == 0, lower = (lane & J) == 0;
  const bool take_min = lower == up;
  const bool other_less = other < key;
  key = (take_min == other_less) ? other : key;
  if constexpr (J > 1) return bitonic_step64<K, J / 2>(key, lane);
  else return key;
}
template <int K = 2>
__device__ __forceinline__ unsigned long long wave_bitonic_sort64(unsigned long long key, int lane) {
  key = bitonic_step64<K, K / 2>(key, lane);
  if constexpr (K < 64) return wave_bitonic_sort64<2 * K>(key, lane);
  else return key;
}

// ---- LDS bitonic sort with its short-distance steps in registers (round 6) ---------------------------------------------------------------------
// Lists beyond the rank sort's reach (1024 < len <= CAP keys in LDS).  The classic network, n = len rounded up to a power of two, padding = +inf:
// stage k = 2 .. n, steps j = k/2 .. 1, comparator (i, i ^ j) ascending where (i & k) == 0.  Every step at a distance below 64 stays inside a block
// of 64 consecutive keys -- one wave's registers (wave_bitonic_sort64's DPP / permlane exchanges): stages 2 .. 64 are one pass over the array, and
// stage k > 64 is log2(k) - 6 passes through LDS + one register pass.  n = 2048: 21 passes over the keys instead of the 66 of the all-LDS network
// (bitonic_any_len, still the in-place global-memory form).  Measured at 1080p / 3 M Gaussians (tiles of ~1100 pairs): the sort kernel's time does not
// move -- nor does it with this network taking over from the rank sort at 512, 256 or 128 keys: at these lengths both cost ~6 us of a CU per tile,
// vector instructions, 200 of the kernel's 400 us.  (A descending block is sorted as the ascending sort of the complemented keys.)
__device__ __forceinline__ void bitonic_lds_regs(unsigned long long* sk, int len, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  int n = 64;
  while (n < len) n <<= 1;
  for (int i = len + tid; i < n; i += 256) sk[i] = ~0ull;
  __syncthreads();
  const int nblk = n >> 6;
  for (int blk = wv; blk < nblk; blk += 4) {
    const bool up = (blk & 1) == 0;      // stage 64's direction
    unsigned long long key = sk[blk * 64 + lane];
    key = up ? key : ~key;
    key = wave_bitonic_sort64(key, lane);
    sk[blk * 64 + lane] = up ? key : ~key;
  }
  __syncthreads();
  for (int k = 128; k <= n; k <<= 1) {
    for (int j = k >> 1; j >= 64; j >>= 1) {
      for (int p = tid; p < (n >> 1); p += 256) {
        const int i = (p / j) * (2 * j) + (p % j), l = i + j;
        const bool up = (i & k) == 0;
        const unsigned long long a = sk[i], c = sk[l];
        if ((a > c) == up) { sk[i] = c; sk[l] = a; }
      }
      __syncthreads();
    }
    for (int blk = wv; blk < nblk; blk += 4) {
      const bool up = ((blk * 64) & k) == 0;
      unsigned long long key = sk[blk * 64 + lane];
      key = up ? key : ~key;
      key = bitonic_step64<64, 32>(key, lane);      // half-cleaners at distances 32 .. 1, ascending
      sk[blk * 64 + lane] = up ? key : ~key;
    }
    __syncthreads();
  }
}

#define RANK_SORT_MAX 1024  // lists up to this length are rank-sorted (needs 2 * RANK_SORT_MAX <= CAP keys of LDS)

// Handles a tile with lo < len <= CAP in LDS (sk[CAP]); when GLOBAL_TAIL it also sorts len > CAP in place in global
// memory (rare: > 16 K splats on one tile).  After sorting it emits the sixteen block lists and their lengths
// (iv.subcount and sh.run).  Returns false when this tier leaves the tile to another launch (nothing written).
// Every lane of the 256-lane workgroup must call it (barriers inside).
template <int CAP, bool GLOBAL_TAIL>
__device__ __forceinline__ bool sort_tile_body(int tile, int gx, int lo, const GeomView& g, const ImageView& iv, const BinView& b,
                                               uint32_t N_cap, int clean, unsigned long long* sk, SortShared& sh, int ex = 0,
                                               int direct_blocks = 0, uint32_t direct_cap = 0, int slot_bits = DIRECT_SLOT_BITS_MAX, SortEmit* emit = nullptr) {
  const uint32_t slot_mask = (1u << slot_bits) - 1u;
  // emit == nullptr: the scratch sits behind the keys | runs | payloads of the rank sort (3 * RANK_SORT_MAX words of `sk`: the caller's LDS block
  // is at least that + sizeof(SortEmit) -- the fused sort + forward kernel, whose staging memory is; the emission that uses it keeps at most 2048 keys in `sk`)
  if (!emit) emit = (SortEmit*)(sk + 3 * RANK_SORT_MAX);
  uint32_t (*wcnt)[NLIST] = emit->wcnt;
  uint32_t (*pre)[NLIST] = emit->pre;
  uint32_t* run = sh.run;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // direct bins (direct_blocks = number of projection workgroups, 0 = packed bins): the tile's pairs sit in its fixed span,
  // their number is the tile's cursor, and every pair's payload carries its block mask and gradient-record index
  const bool direct = direct_blocks != 0;
  uint32_t start, ulen;
  unsigned long long pre0 = ~0ull, pre1 = ~0ull;   // direct bins: the keys of this wave's first two runs, requested before the count is known
  if (direct) {
    // (the span's position comes from a kernel argument -- the same value the binning kernel left in hdr->bin_cap -- so that
    //  the first keys can be requested together with the tile's count instead of one memory round trip after it; reads inside
    //  the span are always in bounds, entries past the count are discarded below)
    const uint32_t cap = direct_cap;
    start = (uint32_t)tile * cap;
    if ((uint32_t)(wv * 64 + lane) < cap) pre0 = b.keys[start + (uint32_t)(wv * 64 + lane)];
    if ((uint32_t)((wv + 4) * 64 + lane) < cap) pre1 = b.keys[start + (uint32_t)((wv + 4) * 64 + lane)];
    const uint32_t cnt = iv.cursor[tile], seen_max = iv.hdr->max_tile_len;
    ulen = min(cnt, cap);
    if (tid == 0) {
      iv.ranges[tile] = cnt;                     // (tile_span clamps to the span)
      if (cnt > cap) iv.hdr->overflow = 1u;
      if (cnt > seen_max) atomicMax(&iv.hdr->max_tile_len, cnt);     // (rare: the maximum is sticky)
    }
    if (tile == 0) {
      // N = pairs touched = sum of the projection workgroups' totals (g.block_tiles[0..direct_blocks)); ONE writer -- 1200
      // same-address atomics would serialise in L2 for ~20 us
      uint32_t x = 0;
      for (int i = tid; i < direct_blocks; i += 256) x += g.block_tiles[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0) sh.scan_tot[wv] = x;
      __syncthreads();
      if (tid == 0) {
        const uint32_t n = sh.scan_tot[0] + sh.scan_tot[1] + sh.scan_tot[2] + sh.scan_tot[3];
        iv.hdr->num_rendered = n;
        iv.hdr->max_num_rendered = max(iv.hdr->max_num_rendered, n);
      }
    }
  } else {
    start = min(iv.ranges[tile], N_cap);
    ulen = min(iv.ranges[tile + 1], N_cap) - start;
    if (clean && tid == 0) { iv.tile_count[tile] = 0; iv.cursor[tile] = 0; }   // scatter_scan_kernel's counters stay zero
  }
  const int len = (int)ulen;
  if (tid == 0) { sh.start = start; sh.len = ulen; }
  if (lo == 0 && len == 0) {
    if (tid < NLIST) { iv.subcount[NLIST * tile + tid] = 0; run[tid] = 0; }
    return true;
  }
  if (len <= lo) return false;
  if (!GLOBAL_TAIL && len > CAP) return false;
  unsigned long long* gk = b.keys + start;
  const bool in_lds = len <= CAP;
  if (in_lds && (ex & 8)) {   // (probe builds, bit 3: no sort; `ex` is the constant 0 in the product build)
    for (int i = tid; i < len; i += 256) sk[i] = gk[i];
    __syncthreads();
  } else if (in_lds && len <= RANK_SORT_MAX) {
    // Run sort + rank merge (keys are unique).  (1) every wave bitonic-sorts 64-key runs in registers (21 compare-exchange
    // steps through the LDS crossbar, no barriers); (2) the rank of a key = its position in its own run + its lower bound
    // in every other run (7-step binary searches, four runs interleaved), and the key goes straight to its final slot.
    // ~250 instructions per key-lane instead of 3 x len for the all-pairs rank sort (870 at the SLAM average of 290).
    unsigned long long* sk2 = sk + RANK_SORT_MAX;
    const int nruns = (len + 63) >> 6;
    for (int r = wv; r < nruns; r += 4) {
      const int i = r * 64 + lane;
      unsigned long long key = ~0ull;                      // padding sorts to the end of the last run
      if (i < len) key = (direct && r == wv) ? pre0 : ((direct && r == wv + 4) ? pre1 : gk[i]);
      key = wave_bitonic_sort64(key, lane);
      sk2[i] = key;
    }
    __syncthreads();
    for (int i = tid; i < len; i += 256) {
      const unsigned long long mine = sk2[i];
      // direct bins: the entry's payload is requested now and lands while the rank is being computed
      const unsigned long long pl = direct ? b.payload[start + ((uint32_t)mine & slot_mask)] : 0ull;
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
      if (direct) sk[2 * RANK_SORT_MAX + rank] = pl;     // (fused kernel: 24 KB of LDS = keys | runs | payloads)
    }
    __syncthreads();
  } else if (in_lds) {
    for (int i = tid; i < len; i += 256) sk[i] = gk[i];
    __syncthreads();
    bitonic_lds_regs(sk, len, tid);      // (the padding stays inside sk[CAP]: CAP is a power of two)
  } else {
    __syncthreads();
    bitonic_any_len([&](int i) -> unsigned long long& { return gk[i]; }, len, tid, 256);
  }
  if (ex & 4) return true;   // (probe builds, bit 2: no emission)
  if (direct && in_lds && len <= RANK_SORT_MAX && !(ex & 8)) {
    // ---- direct bins: every wave builds the four lists of ITS OWN 8x8 sub-tile (the ones it composites) from the sorted ids
    // and payloads in LDS: ballots and running counts inside the wave, no barriers, no cross-wave prefix
    if (tid == 0) iv.cursor[tile] = 0;       // (every wave has read the count: the barrier after the rank merge is behind us)
    const unsigned long long* spl = sk + 2 * RANK_SORT_MAX;
    uint2* sub = b.sublist + (size_t)NLIST * start;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t cnt[4] = {0u, 0u, 0u, 0u};
    for (int base = 0; base < len; base += 64) {
      const int i = base + lane;
      const bool have = i < len;
      const unsigned long long pl = have ? spl[i] : 0ull;
      const uint32_t id = have ? ((uint32_t)sk[i] >> slot_bits) : 0u;
      const uint32_t m4 = ((uint32_t)pl >> (4 * wv)) & 0xfu;
      // the bin in SORTED order (block mask | per-tile record << 32) over the keys, which nobody reads again: the backward compositor's per-tile
      // combine walks it and recomputes every entry's list positions (= its block records, addressed by list position: composite.hip)
      if (have && ((base >> 6) & 3) == wv) b.keys[start + (uint32_t)i] = pl;
#pragma unroll
      for (int bq = 0; bq < 4; bq++) {
        const int L = 4 * wv + bq;
        const bool on = (m4 >> bq) & 1u;
        const unsigned long long bal = __ballot(on);
        if (on) sub[(size_t)L * len + cnt[bq] + __popcll(bal & lt)] = make_uint2(id, 0u);
        cnt[bq] += (uint32_t)__popcll(bal);
      }
    }
    if (lane < 4) {
      const uint32_t c = lane == 0 ? cnt[0] : (lane == 1 ? cnt[1] : (lane == 2 ? cnt[2] : cnt[3]));
      iv.subcount[NLIST * tile + 4 * wv + lane] = c;
      run[4 * wv + lane] = c;
    }
    return true;
  }
  // ---- emit the 16 block lists (order preserving) ----
  // (Measured in round 6 and not kept: packed bins of up to 1024 pairs forming their payloads in the rank loop -- the gathers by id in the order of the
  //  64-key runs -- and emitting from LDS like the direct bins: the sort kernel's time at 1080p / 3 M Gaussians does not move, 541 - 549 us either way.
  //  Of its 400 us there, 200 are the sort proper -- vector instructions, ~6 us of a CU per 1100-key tile whichever of the two sorts runs.)
  const int ttx = tile % gx, tty = tile / gx;
  if (tid < NLIST) run[tid] = 0;
  __syncthreads();
  // the binning kernel's counters stay zero (reset only now: every wave has read the count, none can still see the zero)
  if (direct && tid == 0) iv.cursor[tile] = 0;
  uint2* sub = b.sublist + (size_t)NLIST * start;
  for (int base = 0; base < len; base += 256) {
    const int i = base + tid;
    const bool have = i < len;
    uint32_t id = 0, mask = 0;
    if (have && direct) {
      const uint32_t low = (uint32_t)(in_lds ? sk[i] : gk[i]);
      id = low >> slot_bits;
      const unsigned long long pl = b.payload[start + (low & slot_mask)];
      mask = (uint32_t)(pl & 0xffffu);
      b.keys[start + (uint32_t)i] = pl;      // the bin in sorted order (mask | per-tile record << 32): see the fast path above
    } else if (have) {
      id = (uint32_t)(in_lds ? sk[i] : gk[i]);
      const float4* sp = (const float4*)(g.splat + (size_t)id * SPLAT_F);
      const float4 A = sp[0], B = sp[1];
      // pair index of (Gaussian, tile) in Gaussian-major order = the pair's per-tile gradient record (a Gaussian's pairs are contiguous: the backward
      // projection reads them as one span)
      const uint32_t r0 = g.rect[(size_t)id * 2], r1 = g.rect[(size_t)id * 2 + 1];
      const int minx = r0 & 0xffff, miny = r0 >> 16, rw = (int)(r1 & 0xffff) - minx;
      const uint32_t pidx = g.block_tiles[id >> 8] + g.tileoff[id] + (uint32_t)((tty - miny) * rw + (ttx - minx));
      const BlkRect br = block_rect(A, B, r0, r1);
      mask = tile_block_mask_in_rect(mask_consts(A, B), ttx, tty, br);
      // the bin in sorted order over the keys, which nobody reads again: block mask | per-tile record << 32.  The backward compositor's per-tile
      // combine walks it and recomputes every entry's list positions (= its block records, addressed by list position: composite.hip) -- every
      // mode since round 6 (the generic entry points used to address block records through a per-entry index into a Gaussian-major array)
      b.keys[start + (uint32_t)i] = (unsigned long long)mask | ((unsigned long long)(pidx < N_cap ? pidx : 0xffffffffu) << 32);
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
#pragma unroll
      for (int L = 0; L < NLIST; L++)
        if ((mask >> L) & 1u) sub[(size_t)L * len + pre[wv][L] + __popcll(bal[L] & lt)] = make_uint2(id, 0u);
    }
    __syncthreads();
    if (tid < NLIST) run[tid] = pre[3][tid] + wcnt[3][tid];
    __syncthreads();
  }
  if (tid < NLIST) iv.subcount[NLIST * tile + tid] = run[tid];
  return true;
}

