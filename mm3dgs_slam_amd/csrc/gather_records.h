// Gather of a Gaussian's gradient records (written by composite_bwd, one per (splat, 4x4 block)), shared by the generic and
// the SLAM backward projection kernels.  Every lane of the workgroup must call it (wave votes inside).
#pragma once
#include "mm3dgs_common.h"
#include "composite_common.h"
#include "tile_mask.h"

// NF4: float4s per record that carry data (2 or 3); RECF: record stride in floats (0: run-time stride `recf`, the generic path's
// 6 + C); NTHREADS: workgroup size.
// Inputs per lane: area (tiles in the splat's rectangle, 0 = culled), goff (first pair index), r0/r1 (tile rectangle),
// sA/sB (first 32 bytes of the splat record), bblk + boff (first record).  Output: acc0..acc2 = sum of the records.
// DIRECT (direct bins, which have no Gaussian-major pair index to address submask[] with): the block masks of a small splat
// arrive as one 64-bit word (m64: written per Gaussian by the binning kernel), the blocks of a big splat are re-tested with
// the very rule the lists were built with (tile_mask.h).
template <int NF4, int RECF_T, int NTHREADS, bool DIRECT = false>
__device__ __forceinline__ void gather_records(int area, uint32_t goff, uint32_t r0, uint32_t r1, const float4& sA, const float4& sB,
                                               uint32_t rec_first, const float* __restrict__ dsub, const BinView& bn, uint32_t N_cap,
                                               float4& acc0, float4& acc1, float4& acc2, unsigned long long m64 = 0ull, int recf = 0) {
  const int RECF = RECF_T ? RECF_T : recf;
  constexpr bool TRACK = NF4 == 2;
  {
    // Gradient records: one per (splat, 4x4 block), dense and contiguous per Gaussian (row-major over its block rectangle,
    // mm3dgs_common.h).  Validity comes from the 16-bit block masks of its (Gaussian, tile) pairs: each lane walks ITS OWN
    // set bits, eight at a time (up to 24 independent 16-byte loads in flight: the kernel is latency bound at ~2.4 waves
    // per SIMD); a lane that has run out reads record 0 (one shared, cached line) and discards it.  Ascending bit order =
    // fixed summation order -> deterministic.  10 floats at a 48-B stride (mapping) / 7 floats at a 32-B stride (tracking).
    BlkRect br = {0, 0, 0, 0};
    uint32_t rec0 = 0;
    int tminx = 0, tminy = 0, tw = 1;
    MaskConsts mc;
    mc.cx = mc.cy = mc.hx = mc.hy = mc.r2 = 0.f; mc.mode = 0;
    if (area > 0) {
      br = block_rect(sA, sB, r0, r1);
      rec0 = rec_first;
      tminx = r0 & 0xffff; tminy = r0 >> 16; tw = max((int)(r1 & 0xffff) - tminx, 1);
      if (DIRECT) {
        mc = mask_consts(sA, sB);
        // records beyond the scratch capacity were never written (the forward flagged the overflow): read nothing
        if ((size_t)rec0 + (size_t)br.bw * br.bh > (size_t)NLIST * N_cap) area = 0;
      }
    }
    // M: block masks of up to four pairs; (ox[p], oy[p]) = block coordinates of pair p's tile relative to the block rectangle
    auto drain = [&](unsigned long long M, const int (&ox)[4], const int (&oy)[4], int bw, uint32_t base) {
      constexpr int UB = 8;   // records in flight per lane
      while (__ballot(M != 0ull) != 0ull) {
        float4 a[UB], b[UB], c[UB];
        bool on[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
          on[u] = M != 0ull;
          const int bit = on[u] ? __ffsll((long long)M) - 1 : 0;
          M &= M - 1ull;
          const int pq = bit >> 4, Lb = bit & 15;
          const int oxp = pq == 0 ? ox[0] : (pq == 1 ? ox[1] : (pq == 2 ? ox[2] : ox[3]));
          const int oyp = pq == 0 ? oy[0] : (pq == 1 ? oy[1] : (pq == 2 ? oy[2] : oy[3]));
          const int bx = oxp + ((Lb >> 2) & 1) * 2 + (Lb & 1), by = oyp + (Lb >> 3) * 2 + ((Lb >> 1) & 1);
          const float* r = dsub + (on[u] ? (size_t)(base + (uint32_t)(by * bw + bx)) * RECF : (size_t)0);
          a[u] = ld4u(r); b[u] = ld4u(r + 4);      // (packed records: the lanes past a record's own floats are never used)
          c[u] = TRACK ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4u(r + 8);
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
          acc0.x += on[u] ? a[u].x : 0.f; acc0.y += on[u] ? a[u].y : 0.f; acc0.z += on[u] ? a[u].z : 0.f; acc0.w += on[u] ? a[u].w : 0.f;
          acc1.x += on[u] ? b[u].x : 0.f; acc1.y += on[u] ? b[u].y : 0.f; acc1.z += on[u] ? b[u].z : 0.f; acc1.w += on[u] ? b[u].w : 0.f;
          if (!TRACK) {
            acc2.x += on[u] ? c[u].x : 0.f; acc2.y += on[u] ? c[u].y : 0.f; acc2.z += on[u] ? c[u].z : 0.f; acc2.w += on[u] ? c[u].w : 0.f;
          }
        }
      }
    };
    auto mask_of = [&](uint32_t gi, bool have) -> unsigned long long { return (have && gi < N_cap) ? (unsigned long long)bn.submask[gi] : 0ull; };
    // Small splats (<= 4 tiles, <= 20 blocks: nearly all of a SLAM map; thresholds measured) are summed by their own lane; the rest go through
    // the wave's flat work list below -- one lane walking the hundreds of records of a 30-pixel splat alone was a
    // 60 us tail on a 20 us kernel once a few dozen such splats had grown in the map.
    const int nblk = br.bw * br.bh;
    const bool isbig = area > 4 || nblk > 20;
    if (!isbig) {
      // four pairs (64 mask bits) per round; the first round covers almost every SLAM splat
      int tx = 0, ty = 0;   // tile of pair k0 inside the splat's tile rectangle (row-major, width tw)
      for (int k0 = 0; __ballot(k0 < area) != 0ull; k0 += 4) {
        unsigned long long M = 0ull;
        int ox[4], oy[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (DIRECT) M = m64;          // (a small splat has at most four tiles: one round)
          else M |= mask_of(goff + (uint32_t)(k0 + k), k0 + k < area) << (16 * k);
          ox[k] = (tminx + tx) * 4 - br.bx0; oy[k] = (tminy + ty) * 4 - br.by0;
          if (++tx == tw) { tx = 0; ty++; }
        }
        drain(M, ox, oy, br.bw, rec0);
      }
    }
    // ---- flat work list of the wave's big splats: item = one 4x4 block of a big splat's block rectangle (its records are
    // contiguous: record = rec0 + position).  The S items of the wave are cut into 64 equal contiguous spans, one per lane;
    // a lane sums the records of its span (validity from the block masks) and adds the partial sums of each owner it
    // crosses to that owner's accumulator in LDS.  Same code, same data, same lane order every run -> deterministic.
    const int lane = threadIdx.x & 63, wvq = threadIdx.x >> 6;
    if (__ballot(isbig && area > 0) != 0ull) {
      __shared__ int s_par[NTHREADS / 64][64][DIRECT ? 15 : 9];      // [8]: unused; [9..14]: mask constants
      __shared__ uint32_t s_pref[NTHREADS / 64][64];
      __shared__ float s_acc[NTHREADS / 64][64][12];
      const bool own = isbig && area > 0;
      uint32_t incl = own ? (uint32_t)nblk : 0u;
      incl = wave_scan_incl(incl);
      const uint32_t S = __builtin_amdgcn_readlane(incl, 63);
      s_pref[wvq][lane] = incl;
      int* par = s_par[wvq][lane];
      par[0] = (int)rec0; par[1] = br.bw; par[2] = br.bx0; par[3] = br.by0; par[4] = tminx; par[5] = tminy; par[6] = tw; par[7] = (int)goff;
      par[8] = 0;
      if (DIRECT) {
        par[9] = __float_as_int(mc.cx); par[10] = __float_as_int(mc.cy); par[11] = __float_as_int(mc.hx); par[12] = __float_as_int(mc.hy);
        par[13] = __float_as_int(mc.r2); par[14] = mc.mode;
      }
#pragma unroll
      for (int qv = 0; qv < 12; qv++) s_acc[wvq][lane][qv] = 0.f;
      __builtin_amdgcn_wave_barrier();
      const uint32_t cpl = (S + 63u) / 64u;                       // items per lane
      const uint32_t i0 = min(S, (uint32_t)lane * cpl), i1 = min(S, i0 + cpl);
      int owner = 0;
      {   // first owner: smallest o with pref[o] > i0
        int lo = 0, hi = 63;
#pragma unroll
        for (int st = 0; st < 6; st++) {
          const int mid = (lo + hi) >> 1;
          if (s_pref[wvq][mid] > i0) hi = mid; else lo = mid + 1;
        }
        owner = min(lo, 63);
      }
      float pa[12];
#pragma unroll
      for (int qv = 0; qv < 12; qv++) pa[qv] = 0.f;
      int cur = owner;
      auto flush = [&](int o) {
#pragma unroll
        for (int qv = 0; qv < (TRACK ? 8 : 12); qv++) atomicAdd(&s_acc[wvq][o][qv], pa[qv]);
#pragma unroll
        for (int qv = 0; qv < 12; qv++) pa[qv] = 0.f;
      };
      constexpr int UF = 4;                                         // items in flight per lane
      for (uint32_t t0 = 0; t0 < cpl; t0 += UF) {                  // wave-uniform trip count
        int ow[UF];
        uint32_t recq[UF];
        uint32_t mk[UF];
        int Lq[UF];
        bool have[UF];
#pragma unroll
        for (int u = 0; u < UF; u++) {
          const uint32_t i = i0 + t0 + (uint32_t)u;
          have[u] = i < i1;
          while (have[u] && owner < 63 && i >= s_pref[wvq][owner]) owner++;
          ow[u] = owner;
          const int* pp = s_par[wvq][owner];
          const uint32_t excl = owner ? s_pref[wvq][owner - 1] : 0u;
          const int j = have[u] ? (int)(i - excl) : 0;
          const int bw = max(pp[1], 1);
          int by = (int)(((float)j + 0.5f) / (float)bw);
          if (by * bw > j) by--;
          if ((by + 1) * bw <= j) by++;
          const int bx = j - by * bw;
          const int ax = pp[2] + bx, ay = pp[3] + by;                 // global block coordinates
          const int k = ((ay >> 2) - pp[5]) * pp[6] + ((ax >> 2) - pp[4]);
          Lq[u] = 4 * ((((ay >> 1) & 1) * 2) + ((ax >> 1) & 1)) + (ay & 1) * 2 + (ax & 1);
          recq[u] = (uint32_t)pp[0] + (uint32_t)j;
          if (DIRECT) {
            MaskConsts om;
            om.cx = __int_as_float(pp[9]); om.cy = __int_as_float(pp[10]); om.hx = __int_as_float(pp[11]); om.hy = __int_as_float(pp[12]);
            om.r2 = __int_as_float(pp[13]); om.mode = pp[14];
            mk[u] = (have[u] && block_listed(om, ax, ay)) ? (1u << Lq[u]) : 0u;
          } else {
            const uint32_t gi = (uint32_t)pp[7] + (uint32_t)k;
            mk[u] = (have[u] && gi < N_cap) ? (uint32_t)bn.submask[gi] : 0u;
          }
        }
        float4 a[UF], b4[UF], c4[UF];
        bool on[UF];
#pragma unroll
        for (int u = 0; u < UF; u++) {
          on[u] = (mk[u] >> Lq[u]) & 1u;
          const float* r = dsub + (on[u] ? (size_t)recq[u] * RECF : (size_t)0);
          a[u] = ld4u(r); b4[u] = ld4u(r + 4);
          if (!TRACK) c4[u] = ld4u(r + 8);
        }
#pragma unroll
        for (int u = 0; u < UF; u++) {
          if (have[u] && ow[u] != cur) { flush(cur); cur = ow[u]; }
          if (on[u]) {
            pa[0] += a[u].x; pa[1] += a[u].y; pa[2] += a[u].z; pa[3] += a[u].w;
            pa[4] += b4[u].x; pa[5] += b4[u].y; pa[6] += b4[u].z; pa[7] += b4[u].w;
            if (!TRACK) { pa[8] += c4[u].x; pa[9] += c4[u].y; pa[10] += c4[u].z; pa[11] += c4[u].w; }
          }
        }
      }
      if (i0 < i1) flush(cur);
      __builtin_amdgcn_wave_barrier();
      if (own) {
        const float* sa = s_acc[wvq][lane];
        acc0 = make_float4(sa[0], sa[1], sa[2], sa[3]); acc1 = make_float4(sa[4], sa[5], sa[6], sa[7]);
        acc2 = make_float4(sa[8], sa[9], sa[10], sa[11]);
      }
    }
  }
}
