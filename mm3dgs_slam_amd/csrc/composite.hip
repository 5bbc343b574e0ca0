// Alpha compositing (forward) and its per-pixel reverse traversal (backward) for gfx950 (wave64).
//
// Work decomposition: one 256-lane workgroup per 16x16 tile, ONE WAVE PER 8x8 SUB-TILE, ONE 16-LANE ROW PER 4x4 BLOCK --
// fully autonomous waves: no workgroup barriers, no atomics.  Each row walks the depth-ordered list of ITS block (built by
// sort_tile_body: exactly the tile's list minus the splats whose { alpha >= 1/255 } bound cannot reach the block, so the
// result equals the reference rule applied to every (pixel, splat) pair of the tile):
//   alpha = min(0.99, o * exp(power)); skip power > 0 or alpha < 1/255; stop before the splat that would push T
//   below 1e-4; out = sum c alpha T + T_final * bg        (SURVEY.md Appendix A).
// A wave step therefore evaluates up to four different splats (SLAM splats cover ~40 pixels; with one list per 8x8
// sub-tile 85 % of the lanes of a step were outside the splat).  The lists are consumed 16 entries per row at a time
// through a software pipeline: {id, record} entries are fetched two chunks ahead, the 48-byte splat records one chunk ahead
// (one record per lane), so the dependent gathers are in flight while the current chunk is composited.  A chunk is parked
// in a wave-private LDS slice (field-major) and read row-uniformly one splat at a time (the next splat's record is
// prefetched into registers while the current one is evaluated).  In the SLAM path the same workgroup first sorts its
// tile (sort_composite_fwd_kernel).
//
// Backward: the per-lane gradient terms of one splat are reduced over the 16-lane row with DPP butterflies (generic:
// 6 + C values, halving; SLAM modes: separable moments, 10 / 7 floats) and the row -- the only writer of that (block, splat)
// record -- stores it with one plain store.  preprocess_bwd later sums a Gaussian's records (dense, contiguous) in a
// fixed order, which makes the whole backward deterministic.
#include <type_traits>
#include "mm3dgs_common.h"
#include "sort_tile.h"
#include "fused_api.h"
#include "loss_pixel.h"
#include "loss_tile.h"

// Staging of splat records for the row-uniform LDS reads: a 16-lane row keeps its 16 entries at [row * STG_ROW, +16).  The odd
// row stride puts the four rows of a wave -- which read four DIFFERENT addresses in one ds_read_b128 -- on different banks
// (with a stride of 16 float4s = 256 B all four hit the same four banks: SQ_LDS_BANK_CONFLICT was 35-47 % of the LDS cycles).
#define STG_ROW 17
#define STG_N (4 * STG_ROW)

#include "composite_common.h"

// Forward compositing of one tile by a 256-lane workgroup.  stg: [buffer][wave][field A|B|C][row * 16 + entry] staging
// (24 KB of LDS); counts_lds: the sixteen list lengths in LDS when the caller has just produced them (fused kernel), else
// nullptr (read from image_state).
template <int C>
__device__ __forceinline__ void composite_fwd_body(int tile, const CamDev& cam, const GeomView& g, const ImageView& iv, const BinView& b,
                                                   uint32_t N_cap, float* __restrict__ out, float4 (*stg)[4][3][STG_N],
                                                   const uint32_t* counts_lds, const TrackLoss* tl = nullptr,
                                                   double (*red)[12] = nullptr, const SortShared* span = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // wave = 8x8 sub-tile wv of the tile; 16-lane row = 4x4 block `row` of the sub-tile, walking its own list
  const int row = lane >> 4, q = lane & 15;
  const int slane = row * STG_ROW + q;     // this lane's slot in the staging buffers
  const int px = (tile % cam.gx) * TILE + (wv & 1) * 8 + (row & 1) * 4 + (q & 3);
  const int py = (tile / cam.gx) * TILE + (wv >> 1) * 8 + (row >> 1) * 4 + (q >> 2);
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  uint32_t start, len;
  if (span) { start = span->start; len = span->len; }        // fused with the sort: the same workgroup just determined the bin
  else tile_span(iv, tile, N_cap, start, len);
  const int L = 4 * wv + row;
  const uint32_t count = len ? min(counts_lds ? counts_lds[L] : iv.subcount[NLIST * tile + L], len) : 0u;   // this row's list length
  uint32_t maxcount = count;
  maxcount = max(maxcount, (uint32_t)__builtin_amdgcn_readlane((int)count, 16));
  maxcount = max(maxcount, (uint32_t)__builtin_amdgcn_readlane((int)count, 32));
  maxcount = max(maxcount, (uint32_t)__builtin_amdgcn_readlane((int)count, 48));
  maxcount = max((uint32_t)__builtin_amdgcn_readlane((int)count, 0), maxcount);
  maxcount = __builtin_amdgcn_readfirstlane(maxcount);
  const uint2* __restrict__ list = b.sublist + (size_t)NLIST * start + (size_t)L * len;

  // stg = [buffer][wave][field A|B|C][row * 16 + entry]: lane-contiguous (conflict-free) writes, and ONE address register
  // per splat for the row-uniform reads (fields are a constant 1 KB apart -> immediate offsets)
  constexpr uint32_t CH = 16;   // list entries staged per row and chunk

  float Tr = 1.f;
  float acc[C];
#pragma unroll
  for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
  uint32_t last_contributor = 0;
  bool done = !inside;
  uint32_t n_iter = 0;

  // pipeline prologue: chunk 0 parked in LDS buffer 0, ids of chunk 1 in registers
  {
    const uint32_t id0 = (uint32_t)q < count ? list[q].x : 0u;
    const SplatRec r0 = load_rec<C>(g.splat, id0, (uint32_t)q < count);
    stg[0][wv][0][slane] = r0.A;
    stg[0][wv][1][slane] = r0.B;
    if (C > 2) stg[0][wv][2][slane] = r0.C;
  }
  uint32_t id_nxt = CH + q < count ? list[CH + q].x : 0u;
  int cur = 0;

  const uint32_t mean_steps = wave_mean_steps(cam, iv);
  wave_prio_by_steps(maxcount, mean_steps);
  for (uint32_t base = 0; base < maxcount; base += CH, cur ^= 1) {
    // issue the gathers for the following chunks before touching this one; they land while it is composited
    const SplatRec rec_n = load_rec<C>(g.splat, id_nxt, base + CH + q < count);
    const uint32_t id_nn = base + 2 * CH + q < count ? list[base + 2 * CH + q].x : 0u;
    const float4 (*wS)[STG_N] = stg[cur][wv];
    const int r16 = row * STG_ROW;
    __builtin_amdgcn_wave_barrier();
    const int cnt = __builtin_amdgcn_readfirstlane((int)min(CH, maxcount - base));
    auto splat_fwd = [&](const float4& A, const float4& B, const float4& Cc, const int j) {
      n_iter++;
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float power = splat_power(dx, dy, A.z, A.w, B.x);
      const float alpha = fminf(0.99f, B.y * SPLAT_EXP(power));
      const bool ok = !done && (base + (uint32_t)j < count) && !(power > 0.f) && !(alpha < ALPHA_MIN);
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
      last_contributor = contrib ? base + (uint32_t)j + 1u : last_contributor;
    };
    // two register sets used alternately: the next splat's LDS reads are in flight while the current one is evaluated
    float4 A0 = wS[0][r16], B0 = wS[1][r16], C0 = wS[2][r16];
    for (int j = 0; j < cnt; j += 2) {
      const int j1 = j + 1 < cnt ? j + 1 : j;
      const float4 A1 = wS[0][r16 + j1], B1 = wS[1][r16 + j1], C1 = wS[2][r16 + j1];
      splat_fwd(A0, B0, C0, j);
      if (j + 1 < cnt) {
        const int j2 = j + 2 < cnt ? j + 2 : j1;
        A0 = wS[0][r16 + j2]; B0 = wS[1][r16 + j2]; C0 = wS[2][r16 + j2];
        splat_fwd(A1, B1, C1, j1);
      }
    }
    if (__ballot(!done && base + CH < count) == 0ull) break;
    stg[cur ^ 1][wv][0][slane] = rec_n.A;
    stg[cur ^ 1][wv][1][slane] = rec_n.B;
    if (C > 2) stg[cur ^ 1][wv][2][slane] = rec_n.C;
    id_nxt = id_nn;
  }
  wave_prio_reset();
  if (cam.stats && lane == 0) atomicAdd(&iv.hdr->fwd_wave_iters, n_iter);
  float fin[C];
#pragma unroll
  for (int ch = 0; ch < C; ch++) fin[ch] = acc[ch] + ((ch < 3 || cam.bg_extras) ? Tr * cam.bg[ch % 3] : 0.f);   // (SLAM bundle: the depth pass is composited over bg too, slam/renderer.py:207-214)
  const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
  if (inside) {
    iv.final_T[pix] = Tr;
    iv.n_contrib[pix] = last_contributor;
#pragma unroll
    for (int ch = 0; ch < C; ch++) out[ch * HW + pix] = fin[ch];
  }
  if constexpr (C == 6) {
    if (tl) {
      // tracking loss folded in: this tile's row of partial sums (what loss_reduce_kernel writes for the same 16x16 tile)
      double ls[12];
#pragma unroll
      for (int k = 0; k < 12; k++) ls[k] = 0.0;
      if (inside) {
        const float rgb[3] = {fin[0], fin[1], fin[2]};
        const float g3[3] = {tl->gt[pix], tl->gt[HW + pix], tl->gt[2 * HW + pix]};
        loss_px_sums(tl->cfg, rgb, fin[4], fin[3], g3, tl->cfg.w_pearson != 0.f ? tl->ref[pix] : 0.f, ls);
      }
      block_sums<12>(ls, red, pearson_double_cols(tl->cfg));
      if (threadIdx.x == 0) {
        double* rowp = tl->partial + (size_t)tile * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) rowp[k] = ls[k];
      }
    }
  }
}

template <int C>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, float* __restrict__ out) {
  __shared__ float4 stg[2][4][3][STG_N];
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T, cam.tilemap);
  if (tile >= T) return;
  composite_fwd_body<C>(tile, cam, g, iv, b, N_cap, out, stg, nullptr);
}

// Sort + forward compositing of a tile in ONE launch: both are one-workgroup-per-tile, so the tile's lists go from the
// sorting phase to the compositing phase of the same workgroup (through global memory -- the backward pass reads them
// again -- ordered by the barrier).  One launch gap less per render, and the sort phase (memory latency, half-idle VALU)
// of one workgroup overlaps the compositing phase (VALU bound) of its neighbours on the CU.  The key array and the
// staging buffers share LDS.  Lists longer than 2048 splats take sort_tile_body's global-memory path (slow, correct).
template <int C>
__global__ void __launch_bounds__(256)
sort_composite_fwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, float* __restrict__ out, int clean, int has_tl,
                          TrackLoss tl, int direct_blocks, uint32_t direct_cap, int slot_bits) {
  __shared__ __align__(16) unsigned char smem[sizeof(float4) * 2 * 4 * 3 * STG_N];   // 24 KB >= 2048 keys (16 KB)
  __shared__ SortShared sh;
  __shared__ double red[4][12];
  static_assert(sizeof(smem) >= 2048 * sizeof(unsigned long long), "LDS union too small for the key array");
  // (round 6: the emission's per-wave counters live behind the sort's words in the staging memory instead of in SortShared -- 26.6 KB of LDS instead of
  //  27.1: SIX workgroups per CU (160 KB in 1280-byte granules) where the grid has more than one round of them: configs[3], 3225 tiles)
  static_assert(sizeof(smem) >= 3 * RANK_SORT_MAX * sizeof(unsigned long long) + sizeof(SortEmit), "LDS union too small for the sort's words and the emission's counters");
  const int T = cam.gx * cam.gy;
  const int tile = slam_tile(cam, iv, blockIdx.x, T);
  if (tile >= T) return;
  sort_tile_body<2048, true>(tile, cam.gx, 0, g, iv, b, N_cap, clean, (unsigned long long*)smem, sh, PROBE_WORD(cam), direct_blocks, direct_cap, slot_bits);
  if (PROBE(cam, 1)) return;   // (-DMM3DGS_PROBES builds only: sort phase alone, timing)
  __syncthreads();   // lists (global) and their lengths (sh.run) are visible to the whole workgroup
  composite_fwd_body<C>(tile, cam, g, iv, b, N_cap, out, (float4 (*)[4][3][STG_N])smem, sh.run, has_tl ? &tl : nullptr, red, &sh);

}

// ---- multi-value wave reduction ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_all(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
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
    // no fma contraction here: v[i] is usually a product, and  fma(a, b, dpp(a*b))  costs mul + mov_dpp + fma where
    // (a*b) + dpp(a*b)  is mul + ONE add with a DPP operand
#pragma clang fp contract(off)
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

#define ROW_ROR12 0x12C

// Separable reduction of the SLAM modes over ONE 16-lane row = one 4x4 pixel block (lane bits 0,1 = x, bits 2,3 = y).
// The geometric moments factor: sum u dx^a dy^b = sum_y dy^b (sum_x u dx^a).  The x direction is reduced first on 8 values
//   a[] = { u, u dx, u dx^2, c0, c1, c2, c3, u }      (c* = colour / depth terms, plain sums; u twice on purpose)
// with a halving butterfly (8 -> 4 -> 2 registers); register i of lane (b0, b1) then holds row-sum idx = i + 2 b1 + 4 b0.
// The y direction works on those two registers and their dy-weighted copies Y_i = X_i * {dy, dy, dy^2} on the lanes
// holding {R0, R1, R0'} (-> My, Mxy, Myy), halved on lane bits 2 and 3.  Result: lanes with b3 = 0 hold the X totals
// (M0 Mx Mxx c0 c1 c2 c3), lanes with b3 = 1 the Y totals, value index idx = b2 + 2 b1 + 4 b0.
template <bool RGB>
struct SepReduce {
  // record position of the value lane q (0..15 within its row) ends up with (-1: nothing to store)
  //   mapping (RGB): [M0 Mx Mxx c0 | c1 c2 c3 My | Mxy Myy];   tracking (!RGB): [M0 Mx Mxx c3 | My Mxy Myy]
  __device__ static __forceinline__ int slot(int q) {
    const int b0 = q & 1, b1 = (q >> 1) & 1, b2 = (q >> 2) & 1, b3 = (q >> 3) & 1;
    const int idx = b2 + 2 * b1 + 4 * b0;
    if (!b3) {
      if (idx <= 2) return idx;
      if (RGB) return idx <= 6 ? idx : -1;
      return idx == 6 ? 3 : -1;
    }
    const int base = RGB ? 7 : 4;
    if (idx == 0) return base;       // My  = sum dy R0
    if (idx == 1) return base + 1;   // Mxy = sum dy R1
    if (idx == 7) return base + 2;   // Myy = sum dy^2 R0'
    return -1;
  }
  // per-lane selectors of the dy weights: Y0 = X0 * dy * m0,  Y1 = X1 * dy * (m1a + m1b * dy)
  __device__ static __forceinline__ void ymult(int q, float& m0, float& m1a, float& m1b) {
    const int b0 = q & 1, b1 = (q >> 1) & 1;
    m0 = (!b0 && !b1) ? 1.f : 0.f;    // register 0 holds idx 0 (R0) there
    m1a = (!b0 && !b1) ? 1.f : 0.f;   // register 1 holds idx 1 (R1) there
    m1b = (b0 && b1) ? 1.f : 0.f;     // register 1 holds idx 7 (R0') there
  }
  __device__ static __forceinline__ float run(float u, float dx, float dy, float c0, float c1, float c2, float c3, int lane, float m0,
                                              float m1a, float m1b) {
#pragma clang fp contract(off)
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    const float udx = u * dx, udxx = udx * dx;
    // bit 0: pairs (i, i + 4)
    float l1[4];
    {
      const float s0 = u + dpp_all<QP_XOR1>(u), s1 = udx + dpp_all<QP_XOR1>(udx), s2 = udxx + dpp_all<QP_XOR1>(udxx);
      const float s6 = c3 + dpp_all<QP_XOR1>(c3);
      if (RGB) {
        const float s3 = c0 + dpp_all<QP_XOR1>(c0), s4 = c1 + dpp_all<QP_XOR1>(c1), s5 = c2 + dpp_all<QP_XOR1>(c2);
        l1[0] = b0 ? s4 : s0; l1[1] = b0 ? s5 : s1;
        l1[3] = b0 ? s0 : s3;     // value 7 is u again (s0), value 3 is c0
      } else {
        l1[0] = s0; l1[1] = s1;   // odd lanes carry junk there: never stored (slot() = -1)
        l1[3] = s0;               // no c0: keep u on both parities (even lanes: junk value 3, never stored)
      }
      l1[2] = b0 ? s6 : s2;
    }
    // bit 1: pairs (i, i + 2)
    float X0, X1;
    {
      const float t0 = l1[0] + dpp_all<QP_XOR2>(l1[0]), t1 = l1[1] + dpp_all<QP_XOR2>(l1[1]);
      const float t2 = l1[2] + dpp_all<QP_XOR2>(l1[2]), t3 = l1[3] + dpp_all<QP_XOR2>(l1[3]);
      X0 = b1 ? t2 : t0;
      X1 = b1 ? t3 : t1;
    }
    // y direction: dy-weighted copies, then bit 2 (partner 4 lanes away: row_ror:n hands lane i the value of lane i - n,
    // measured with tools/ubench/dpp_dir.hip) and bit 3 (partner 8 lanes away)
    const float Y0 = X0 * (dy * m0), Y1 = X1 * (dy * (m1a + m1b * dy));
    const float xa = X0 + dpp_all<ROW_ROR12>(X0), xb = X1 + dpp_all<ROW_ROR4>(X1);
    const float ya = Y0 + dpp_all<ROW_ROR12>(Y0), yb = Y1 + dpp_all<ROW_ROR4>(Y1);
    const float X = b2 ? xb : xa, Y = b2 ? yb : ya;
    const float tX = X + dpp_all<ROW_ROR8>(X), tY = Y + dpp_all<ROW_ROR8>(Y);
    return b3 ? tY : tX;
  }
};

// Round 3: the same separable reduction with the Y direction FIRST and its halving steps done by bank-masked DPP adds.  A DPP bank =
// four consecutive lanes = one pixel row of the block (lane bits 2, 3 = y): `v_add_f32_dpp ... bank_mask` writes only the lanes of the
// selected pixel rows, so "lanes with y even keep value i, lanes with y odd keep value i + 4" is two masked adds into one register
// instead of two full adds and a select (the halving on lane bits 0, 1 needs the selects: a mask cannot tell the lanes of a bank apart).
//   Y stage on 8 (mapping) / 4 (tracking) values  { u, u dy, u dy^2, c0, c1, c2, cz, u }  /  { u, u dy, u dy^2, cz }:
//     partner row y ^ 1 (row_ror:12 = lane + 4 for the even rows, row_ror:4 = lane - 4 for the odd ones), then y ^ 2 (row_ror:8);
//     afterwards pixel row y holds the column sums (over y) of two (one) of the values.
//   X stage: up to three x-weighted copies per row ( 1 | dx | dx^2 ), full butterfly over the four lanes of the bank (quad_perm).
// 27 (mapping) / 18 (tracking) instructions per (row, splat) step instead of 33 / 27.  The masked adds are inline assembly (the compiler's
// DPP combiner does not form them); `s_nop 1` covers the VALU-write -> DPP-read hazard at the head of the block (2 wait states = TWO
// instructions between the write and the DPP read); inside the mapping block every register is DPP-read at least three instructions
// after its last write, the shorter tracking block needs one `s_nop 0` (r0 / r1 had only one instruction between).
// Summation order differs from SepReduce in the last bit only.
template <bool RGB>
struct SepReduce2 {
  // record position of the value lane q ends up with (-1: nothing to store); layouts as SepReduce
  __device__ static __forceinline__ int slot(int q) {
    const int x = q & 3, y = q >> 2;
    if (RGB) {
      // row 0: Mx My Mxy | row 1: c1 c2 - | row 2: Myy c0 - | row 3: cz M0 Mxx      ([M0 Mx Mxx c0 | c1 c2 cz My | Mxy Myy])
      const int t[4][4] = {{1, 7, 8, -1}, {4, 5, -1, -1}, {9, 3, -1, -1}, {6, 0, 2, -1}};
      return t[y][x];
    }
    // row 0: M0 Mx Mxx | row 1: Myy | row 2: My Mxy | row 3: cz                     ([M0 Mx Mxx cz | My Mxy Myy])
    const int t[4][4] = {{0, 1, 2, -1}, {6, -1, -1, -1}, {4, 5, -1, -1}, {3, -1, -1, -1}};
    return t[y][x];
  }
  __device__ static __forceinline__ float run(float u, float dx, float dy, float c0, float c1, float c2, float cz, int lane) {
#pragma clang fp contract(off)
    const int x = lane & 3, y = (lane >> 2) & 3;
    const float udy = u * dy, udyy = udy * dy;
    float t0, t1, t2;
    if (RGB) {
      float r0, r1, r2, r3, s0, s1;
      asm volatile(
          "s_nop 1\n\t"
          "v_add_f32_dpp %0, %6, %6 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"     // rows 0, 2: u
          "v_add_f32_dpp %1, %7, %7 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"     //            u dy
          "v_add_f32_dpp %2, %8, %8 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"     //            u dy^2
          "v_add_f32_dpp %3, %9, %9 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"     //            c0
          "v_add_f32_dpp %0, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"    // rows 1, 3: c1
          "v_add_f32_dpp %1, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"    //            c2
          "v_add_f32_dpp %2, %12, %12 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"    //            cz
          "v_add_f32_dpp %3, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"      //            u (again)
          "v_add_f32_dpp %4, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"      // rows 0, 1 keep r0, r1
          "v_add_f32_dpp %5, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
          "v_add_f32_dpp %4, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"      // rows 2, 3 keep r2, r3
          "v_add_f32_dpp %5, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(s0), "=&v"(s1)
          : "v"(u), "v"(udy), "v"(udyy), "v"(c0), "v"(c1), "v"(c2), "v"(cz));
      // row 0: s0 = U, s1 = U1 | row 1: c1, c2 | row 2: U2, c0 | row 3: cz, U
      const float w0 = y == 0 ? dx : 1.f;
      const float w2 = dx * (y == 0 ? 1.f : dx);
      t0 = s0 * w0;       // U dx | c1 | U2 | cz
      t1 = s1;            // U1   | c2 | c0 | U
      t2 = s1 * w2;       // U1 dx | -  | -  | U dx^2
    } else {
      float r0, r1, sy;
      asm volatile(
          "s_nop 1\n\t"
          "v_add_f32_dpp %0, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"     // rows 0, 2: u
          "v_add_f32_dpp %1, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"     //            u dy
          "v_add_f32_dpp %0, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"      // rows 1, 3: u dy^2
          "v_add_f32_dpp %1, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"      //            cz
          "s_nop 0\n\t"       // r0's second write is TWO instructions back only with this (a DPP read needs 2 wait states = 2 intervening instructions after a VALU write; the hazard recognizer does not see inside inline asm)
          "v_add_f32_dpp %2, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"      // rows 0, 1 keep r0: U | U2
          "v_add_f32_dpp %2, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"      // rows 2, 3 keep r1: U1 | cz
          : "=&v"(r0), "=&v"(r1), "=&v"(sy)
          : "v"(u), "v"(udy), "v"(udyy), "v"(cz));
      t0 = sy;            // U | U2 | U1 | cz
      t1 = sy * dx;       // U dx | - | U1 dx | -
      t2 = t1 * dx;       // U dx^2 | - | - | -
    }
    // X stage: full butterfly over the four lanes of every pixel row (six fused DPP adds: left to the compiler, the second level came out
    // as v_mov_b32_dpp + v_add_f32 inside exec-masked branches), then the lane picks the total it stores
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(t0), "+v"(t1), "+v"(t2));
    const float t01 = x == 0 ? t0 : t1;
    return x >= 2 ? t2 : t01;
  }
};
#ifdef MM3DGS_OLD_REDUCE
#define SEP_REDUCE2 0
#else
#define SEP_REDUCE2 1
#endif

// Record written per (4x4 block, splat); the moments m = sum_p u_p (1, dx, dy, dx^2, dx dy, dy^2) of u = dL/dG * G over the
// block's pixels give d/dxy and d/dconic with the splat's conic (preprocess_bwd).
// MODE 0: generic, record = [m_x m_y m_xx m_xy m_yy | dopacity | dcolour(C)].
// MODE 1: SLAM mapping (C = 6, colours = rgb | z 1 z^2): [M0 Mx Mxx c0 | c1 c2 cz My | Mxy Myy] with cz = sum w (dL_3 + 2 z dL_5)
//         -- the chain rule of the depth bundle folded into the reduction -- and dopacity = M0 / opacity (10 floats, not 12).
// MODE 2: SLAM tracking: [M0 Mx Mxx cz | My Mxy Myy] at a 32-byte stride: opacity / colour gradients are never consumed.
// (a device function: the tracking loop runs it in the same launch as the sort and the forward compositor, see below)
#define BWD_STG_BYTES (sizeof(float4) * 2 * 4 * 3 * STG_N + sizeof(uint32_t) * 2 * 4 * 64)   // staged splat records + record indices: 26 KB

// ---- round 6: the SLAM modes' backward in TWO PHASES per sub-chunk of list entries ---------------------------------------------------------------
// The one-phase loop (kept as the generic mode's, and under -DMM3DGS_BWD_ONE_PHASE as the SLAM modes' A/B baseline) reduces the ten / seven record
// values over the 16 lanes of a row after EVERY (row, splat) step: 18 / 12 DPP adds, the products that feed them and a 10-lane scatter store -- 33 of
// the step's 64 vector instructions (profiles/r05_isa_budget.txt), although per (pixel, splat) only TWO numbers are new: u = dL/dG G and w = alpha T.
// Here phase 1 (lane = pixel, as before) writes (u, w) of SUB consecutive steps to a wave-private LDS tile, and phase 2 turns the tile around:
// lane = (list entry, part of the 4x4 block), 16 / SUB lanes per entry, each walking its SUB pixels with the entry's centre in registers and the pixels'
// dL from a wave-private LDS table -- plain fmas into ten accumulators, one DPP add per value and part to merge them, and the record leaves as two wide
// stores from the lanes that hold it.  Deterministic (fixed order: x inside a pixel row, rows ascending, parts ascending).  The summation order differs
// from the one-phase reduction's in the last bits.
#ifdef MM3DGS_BWD_ONE_PHASE
#define BWD_TWO_PHASE 0
#else
#define BWD_TWO_PHASE 1
#endif
// (developer timing probes of the two-phase loop, variant builds only -- results INVALID: -DMM3DGS_BWD2_PROBE=<bits>  1: no phase 2 | 2: phase 2 without its
//  record stores | 4: phase 1 without the (u, w) tile writes | 8: phase 2 on constants instead of its tile / table reads)
#ifndef MM3DGS_BWD2_PROBE
#define MM3DGS_BWD2_PROBE 0
#endif
// generic mode, three channels (the RGB pass of the reference's rasterizer: configs[4]): the two-phase loop too (round 6); -DMM3DGS_GEN3_ONE_PHASE keeps its
// one-phase loop (A/B baseline).  Other channel counts of the generic mode stay on the one-phase loop.
#ifdef MM3DGS_GEN3_ONE_PHASE
#define GEN3_TWO_PHASE 0
#else
#define GEN3_TWO_PHASE BWD_TWO_PHASE
#endif
#define BWD_IS_TWO_PHASE(MODE, C) (((MODE) != 0 && BWD_TWO_PHASE) || ((MODE) == 0 && (C) == 3 && GEN3_TWO_PHASE))
#define TP_STRIDE 65      // float2 per step row of the (u, w) tile: 64 lanes + 1 (the SUB lanes of a phase-2 group read SUB different rows at one column: the odd stride spreads them over the banks)
template <int MODE>
struct Bwd2Lds {      // wave-private slice of the workgroup's LDS block (bytes)
  static constexpr int OFF_A = 0;                                  // float4[STG_N]: px py conA conB          (single buffer: the next chunk is parked after the last read of this one)
  static constexpr int OFF_B = OFF_A + STG_N * 16;                 // float4[STG_N]: conC opacity c0 c1
  static constexpr int OFF_CZ = OFF_B + STG_N * 16;                // float2[STG_N]: c2 z                      (z^2 = z * z is recomputed: one rounded product, as the projection stored it)
  static constexpr int OFF_TP = OFF_CZ + STG_N * 8;                // float2[8][TP_STRIDE]: (u, w) of a sub-chunk
  static constexpr int OFF_T5 = OFF_TP + 4 * TP_STRIDE * 8;        // float[64]: dL5 per pixel -- the general instance runs SUB = 4 and keeps its fifth table in the tile's unused half
  static constexpr int OFF_T = OFF_TP + 8 * TP_STRIDE * 8;         // mapping: float4[64] (dL0 dL1 dL2 dL3) per pixel; tracking: float[64] (dL3)
  static constexpr int SLICE = OFF_T + (MODE == 1 ? 64 * 16 : 64 * 4);
};
#define BWD2_BYTES(MODE) (4 * Bwd2Lds<MODE>::SLICE)      // mapping 31616 B, tracking 28544 B: five workgroups per CU (160 KB in 1280-byte granules: at most 32000)
// POSE (tracking, two-phase loop only): the pose chain of fused.hip's pose_chain_record -- phase 2 applies the splat's { Kp, Kq, x } (GeomView.poserec) to
// the block's moments and adds dm (x) [x; 1] to per-lane accumulators; at the end the workgroup writes ONE row of twelve floats, dsub[tile][32], for the
// pose finish.  No gradient record is written, none zeroed, no per-tile combine, and no backward-projection launch follows.
template <int C, int MODE, bool POSE = false>
__device__ __forceinline__ void composite_bwd_body(int tile, const CamDev& cam, const GeomView& g, const ImageView& iv, const BinView& b,
                                                   uint32_t N_cap, const float* __restrict__ dL_dout, float* __restrict__ dsub, int has_tl,
                                                   const TrackLoss& tl, int dl_planes, unsigned char* smem_raw, const SortShared* span = nullptr) {
  static_assert(!POSE || (MODE == 2 && BWD_TWO_PHASE), "the pose chain lives in the tracking mode's two-phase loop");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // wave = 8x8 sub-tile wv of the tile; 16-lane row = 4x4 block `row` of the sub-tile, walking its own list
  const int row = lane >> 4, q = lane & 15;
  const int slane = row * STG_ROW + q;     // this lane's slot in the staging buffers
  const int px = (tile % cam.gx) * TILE + (wv & 1) * 8 + (row & 1) * 4 + (q & 3);
  const int py = (tile / cam.gx) * TILE + (wv >> 1) * 8 + (row >> 1) * 4 + (q >> 2);
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  uint32_t start, len;
  if (span) { start = span->start; len = span->len; }
  else tile_span(iv, tile, N_cap, start, len);
  const int L = 4 * wv + row;
  const uint32_t count = len ? min(span ? span->run[L] : iv.subcount[NLIST * tile + L], len) : 0u;   // this row's list length
  const uint2* __restrict__ list = b.sublist + (size_t)NLIST * start + (size_t)L * len;

  constexpr int NV = MODE == 0 ? 6 + C : (MODE == 1 ? 10 : 7);   // floats per record
  // (the per-tile combine at the end of this function costs 7.6 us of a 77.5 us launch on a static 150 k scene: 4.7 without its record
  //  loads -- barrier, table loads, stores: the tail of every workgroup -- and 2.9 for the loads; 2, 4 or 8 records in flight make no
  //  difference, and requesting a lane's table entries up here costs more than it saves: DESIGN.md section 4)
  static_assert(MODE == 0 || C == 6, "SLAM modes composite the 6-channel bundle");
  constexpr int RECF = MODE == 0 ? GENERIC_RECF(C) : (MODE == 1 ? REC_MAP_F : REC_TRACK_F);   // record stride in floats: packed at the record's real size (composite_common.h; generic: 6 + C)
  // [buffer][wave][field A|B|C][row * 16 + entry] + [buffer][wave][row * 16 + entry] record indices: lane-contiguous
  // (conflict-free) writes, and ONE address register per splat for the row-uniform reads (fields are a constant 1 KB apart ->
  // immediate offsets).  (The caller's LDS block is shared with the scratch of the folded mapping-loss gradient pass, which
  // runs first, and -- in the fused tracking kernel -- with the sort keys and the forward compositor's staging buffers.)
  float4 (*stg)[4][3][STG_N] = (float4 (*)[4][3][STG_N])smem_raw;
  constexpr uint32_t CH = 16;   // list entries staged per row and chunk

  const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
  const float T_final = inside ? iv.final_T[pix] : 0.f;
  const uint32_t last_contributor = inside ? iv.n_contrib[pix] : 0u;
  // nothing behind the deepest contributor of any pixel of the block matters: todo = max over the row
  uint32_t todo = last_contributor;
  todo = max(todo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)todo, QP_XOR1, 0xf, 0xf, true));
  todo = max(todo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)todo, QP_XOR2, 0xf, 0xf, true));
  todo = max(todo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)todo, ROW_ROR4, 0xf, 0xf, true));
  todo = max(todo, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)todo, ROW_ROR8, 0xf, 0xf, true));
  todo = min(todo, count);
  const uint32_t maxtodo = max(max((uint32_t)__builtin_amdgcn_readlane((int)todo, 0), (uint32_t)__builtin_amdgcn_readlane((int)todo, 16)),
                               max((uint32_t)__builtin_amdgcn_readlane((int)todo, 32), (uint32_t)__builtin_amdgcn_readlane((int)todo, 48)));
  float dL[C];
  float bg_dot = 0.f;
  bool dl_done = false;
  if constexpr (MODE == 2) {
    if (has_tl) {
      // tracking loss folded in: dL/d(image) of this pixel from the finished sums (what loss_grad_kernel would have written)
#pragma unroll
      for (int ch = 0; ch < C; ch++) dL[ch] = 0.f;
      if (inside) {
        const float sil = tl.out[4 * HW + pix];
        const bool smask = sil > tl.cfg.sil_thr;
        const float l1s = tl.defer_scale ? tl.cfg.w_l1 / 3.f : loss_l1_scale(tl.cfg, tl.sums);   // deferred: 1/n applied to the pose gradient
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dL[ch] = loss_px_l1_grad(tl.cfg, tl.out[ch * HW + pix], tl.gt[ch * HW + pix], smask, l1s);
        if (tl.cfg.w_pearson != 0.f) dL[3] = loss_px_pearson_grad(tl.cfg, sil, tl.out[3 * HW + pix], tl.ref[pix], tl.sums);
      }
      if (!tl.defer_scale && tile == 0 && threadIdx.x == 0 && tl.loss4) loss_scalars(tl.cfg, tl.sums, HW, tl.loss4);
      dl_done = true;
    }
  }
  if constexpr (MODE == 1) {
    if (has_tl) {
      // mapping loss folded in: the gradient-image pass of loss.hip (adjoint SSIM convolution + L1 sign + Pearson gradient) for
      // this tile, in this workgroup, into registers -- one launch and the dL round trip through HBM less per iteration.  The
      // pass works in raster order (lane = 16 * y + x of the tile); the compositor's lanes pick their pixel up through LDS.
      LossGradSmem& lsm = *(LossGradSmem*)smem_raw;
      float* exch = (float*)(smem_raw + sizeof(LossGradSmem));
      float g4[4];
      bool in_raster;
      if (PROBE(cam, 10)) { g4[0] = g4[1] = g4[2] = g4[3] = 1e-3f; in_raster = true; }      // (probe builds, bit 10: timing without the loss prologue)
      else loss_grad_tile(tl.cfg, tl.out, tl.gt, tl.ref, tl.dmaps, tl.sums, tile, cam.gx, lsm, g4, in_raster);
#pragma unroll
      for (int ch = 0; ch < 4; ch++) exch[ch * 256 + tid] = g4[ch];
      __syncthreads();
      const int lx = (wv & 1) * 8 + (row & 1) * 4 + (q & 3), ly = (wv >> 1) * 8 + (row >> 1) * 4 + (q >> 2);
#pragma unroll
      for (int ch = 0; ch < C; ch++) dL[ch] = ch < 4 ? exch[ch * 256 + ly * 16 + lx] : 0.f;
      __syncthreads();      // the staging buffers reuse this memory
      dl_done = true;
    }
  }
#pragma unroll
  for (int ch = 0; ch < C; ch++) {
    if (!dl_done) dL[ch] = (inside && ch < dl_planes) ? dL_dout[ch * HW + pix] : 0.f;
    if (ch < 3 || cam.bg_extras) bg_dot += cam.bg[ch % 3] * dL[ch];
  }
  const float Tf_bg = T_final * bg_dot;
  float Tr = T_final;
  float behind_dot = 0.f;  // (colour accumulated behind the current list position) . dL

  // entries behind `todo` receive no gradient: their records are zero
  // SLAM modes (round 6): the record of a list entry sits at the entry's own position, NLIST * start + L * len + k -- LIST-major.  A row's walk
  // writes consecutive records (whole cache lines leave the L2 once, instead of one 64-byte write per 40-byte record: the compositor's WRITE_SIZE
  // was 1.6x its record bytes), the zeroed tail is one contiguous span, and the list entries need not carry an index.  The per-tile combine finds a
  // pair's records by recomputing its list positions from the sorted bin (below).  Generic mode: Gaussian-major records, index in the entry.
  const size_t rbase = (size_t)NLIST * start + (size_t)L * len;
  if constexpr (!POSE)
  for (uint32_t e = todo + q; e < count; e += 16) {
    zero_record<NV>(dsub + (rbase + e) * RECF);
  }
  float pacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // POSE: this lane's share of the tile's pose row (lanes 0-7 of a row: dR rows 0, 1; lanes 8-15: dR row 2, dt)
  if constexpr (BWD_IS_TWO_PHASE(MODE, C)) {
  if (maxtodo != 0) {   // wave-uniform (a wave without work still takes part in the workgroup's per-tile combine below)
    using LY = Bwd2Lds<(MODE == 0 ? 1 : MODE)>;      // (generic three-channel instance: the mapping layout -- a float4 of dL per pixel)
    unsigned char* const wbase = smem_raw + (size_t)wv * LY::SLICE;
    float4* const sA = (float4*)(wbase + LY::OFF_A);
    float4* const sB = (float4*)(wbase + LY::OFF_B);
    float2* const sCZ = (float2*)(wbase + LY::OFF_CZ);
    float2* const tp = (float2*)(wbase + LY::OFF_TP);
    float* const t5 = (float*)(wbase + LY::OFF_T5);
    // the pixels' dL for phase 2 (wave-private: a wave only ever reads its own four blocks)
    bool z45_wave = true;      // (generic: no fifth / sixth channel -- the 8-entry instance, with the background term kept)
    if constexpr (MODE == 0) ((float4*)(wbase + LY::OFF_T))[lane] = make_float4(dL[0], dL[1], dL[2], 0.f);
    else {
      if constexpr (MODE == 1) ((float4*)(wbase + LY::OFF_T))[lane] = make_float4(dL[0], dL[1], dL[C > 3 ? 2 : 0], dL[C > 3 ? 3 : 0]);
      else ((float*)(wbase + LY::OFF_T))[lane] = dL[C > 3 ? 3 : 0];
      z45_wave = __ballot(dL[C > 4 ? 4 : 0] != 0.f || dL[C > 5 ? 5 : 0] != 0.f || Tf_bg != 0.f) == 0ull;      // see the one-phase loop below
      if (!z45_wave) t5[lane] = dL[C > 5 ? 5 : 0];
    }
    const uint32_t first_step = todo - min(todo, last_contributor);
    const uint32_t mean_steps_b = wave_mean_steps(cam, iv);
    const float X0 = pxf - (float)(q & 3), Y0 = pyf - (float)(q >> 2);      // pixel centre of the block's corner (exact)
    uint32_t n_visit = 0;
    // chunk c of a row holds its list entries todo-1-(16c+q): entry order == traversal order (back to front); lane q parks entry q and keeps
    // its gradient-record index in a register
    uint32_t my_idx;
    {
      const uint2 e0 = (uint32_t)q < todo ? list[todo - 1u - q] : make_uint2(0u, 0u);
      const SplatRec r0 = load_rec<C>(g.splat, e0.x, (uint32_t)q < todo);
      sA[slane] = r0.A; sB[slane] = r0.B; sCZ[slane] = make_float2(r0.C.x, r0.C.y);
      my_idx = POSE ? e0.x : e0.y;      // (pose chain: the entry's Gaussian, whose { Kp, Kq, x } phase 2 gathers)
    }
    uint2 ent_nxt = CH + q < todo ? list[todo - 1u - (CH + q)] : make_uint2(0u, 0u);
    auto run2 = [&](auto z45_tag) {
      constexpr bool Z45 = decltype(z45_tag)::value;
      constexpr int SUB = Z45 ? 8 : 4;           // entries per phase-2 pass
      constexpr int NL = 16 / SUB;               // lanes per entry = parts of the 4x4 block; each walks SUB pixels = SUB / 4 pixel rows
      const int e = q & (SUB - 1), part = q / SUB;
      const int r17 = row * STG_ROW;
      for (uint32_t base = 0; base < maxtodo; base += CH) {
        const SplatRec rec_n = load_rec<C>(g.splat, ent_nxt.x, base + CH + q < todo);
        const uint2 ent_nn = base + 2 * CH + q < todo ? list[todo - 1u - (base + 2 * CH + q)] : make_uint2(0u, 0u);
        __builtin_amdgcn_wave_barrier();
        const int cnt = __builtin_amdgcn_readfirstlane((int)min(CH, maxtodo - base));
        for (int sub = 0; sub < cnt; sub += SUB) {
          const int cs = min(SUB, cnt - sub);
          // ---- phase 1: lane = pixel; one (row, splat) step = the splat's alpha, the transmittance in front of it, dL/dalpha -> (u, w)
          float2* tpw = tp + lane;
          auto splat_px = [&](const float4& A, const float4& B, const float2& CZ, const int j) {
            const uint32_t step = base + (uint32_t)j;               // wave-uniform
            const bool row_on = step < todo;                        // this row still has an entry at this step
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = splat_power(dx, dy, A.z, A.w, B.x);
            const float G = SPLAT_EXP(power);
            const float alpha = fminf(0.99f, B.y * G);
            const bool valid = row_on && (step >= first_step) && !(power > 0.f) && !(alpha < ALPHA_MIN);
            n_visit++;
            const float a_eff = valid ? alpha : 0.f;
            const float G_eff = valid ? G : 0.f;
            float r;
            if constexpr (MODE == 0) {      // (the reciprocal to within an ulp: see the one-phase loop)
              const float d = 1.f - a_eff, r0 = __builtin_amdgcn_rcpf(d);
              r = fmaf(fmaf(-d, r0, 1.f), r0, r0);
            } else r = __builtin_amdgcn_rcpf(1.f - a_eff);
            Tr *= r;  // transmittance in front of this splat
            const float w = a_eff * Tr;
            // dL/dalpha needs sum_ch (c_ch - behind_ch) dL_ch: ONE running scalar (behind_dot), see the one-phase loop
            float qd = fmaf(B.z, dL[0], 0.f);
            qd = fmaf(B.w, dL[1], qd);
            qd = fmaf(CZ.x, dL[2], qd);
            if constexpr (MODE != 0) qd = fmaf(CZ.y, dL[C > 3 ? 3 : 0], qd);
            if constexpr (!Z45 && MODE != 0) { qd = fmaf(1.f, dL[C > 4 ? 4 : 0], qd); qd = fmaf(CZ.y * CZ.y, dL[C > 5 ? 5 : 0], qd); }
            const float diff = qd - behind_dot;
            behind_dot = fmaf(a_eff, diff, behind_dot);
            const float dLa = (Z45 && MODE != 0) ? diff * Tr : diff * Tr - Tf_bg * r;
            // (generic: u WITHOUT the opacity -- the record wants sum G dL/dalpha beside the moments of o G dL/dalpha; phase 2 scales the moments)
            const float u = MODE == 0 ? dLa * G_eff : B.y * dLa * G_eff;
            if (!(MM3DGS_BWD2_PROBE & 4)) *tpw = make_float2(u, w);
            tpw += TP_STRIDE;
          };
          {
            float4 A0 = sA[r17 + sub], B0 = sB[r17 + sub];
            float2 Z0 = sCZ[r17 + sub];
            for (int j = sub; j < sub + cs; j += 2) {
              const int j1 = j + 1 < sub + cs ? j + 1 : j;
              const float4 A1 = sA[r17 + j1], B1 = sB[r17 + j1];
              const float2 Z1 = sCZ[r17 + j1];
              splat_px(A0, B0, Z0, j);
              if (j + 1 < sub + cs) {
                const int j2 = j + 2 < sub + cs ? j + 2 : j1;
                A0 = sA[r17 + j2]; B0 = sB[r17 + j2]; Z0 = sCZ[r17 + j2];
                splat_px(A1, B1, Z1, j1);
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
          // ---- phase 2: lane = (entry sub + e of the row's chunk, part of the block): the entry's record from the (u, w) tile
          if (!(MM3DGS_BWD2_PROBE & 1)) {
#pragma clang fp contract(off)
            const int ent = sub + e;
            const bool on2 = base + (uint32_t)ent < todo;
            const float2 cxy = *(const float2*)&sA[r17 + ent];
            const uint32_t idx = (uint32_t)__builtin_amdgcn_ds_bpermute((row * 16 + ent) << 2, (int)my_idx);
            float4 k0, k1, k2, k3, k4;
            if constexpr (POSE) {      // requested here, consumed after the moments are merged
              const float4* pr = (const float4*)(g.poserec + (size_t)(on2 ? idx : 0u) * POSEREC_F);
              k0 = pr[0]; k1 = pr[1]; k2 = pr[2]; k3 = pr[3]; k4 = pr[4];
            }
            float dxv[4], dxx[4];
#pragma unroll
            for (int x = 0; x < 4; x++) { dxv[x] = cxy.x - (X0 + (float)x); dxx[x] = dxv[x] * dxv[x]; }
            float M0 = 0.f, Mx = 0.f, Mxx = 0.f, My = 0.f, Mxy = 0.f, Myy = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, cz = 0.f, cz5 = 0.f;
            const float2* tpr = tp + e * TP_STRIDE + row * 16 + part * SUB;
            const int pcol = row * 16 + part * SUB;
#pragma unroll
            for (int yy = 0; yy < SUB / 4; yy++) {
              float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int x = 0; x < 4; x++) {
                const float2 uw = (MM3DGS_BWD2_PROBE & 8) ? make_float2(cxy.x, cxy.y) : tpr[yy * 4 + x];
                s0 = s0 + uw.x;
                s1 = __builtin_fmaf(uw.x, dxv[x], s1);
                s2 = __builtin_fmaf(uw.x, dxx[x], s2);
                if constexpr (MODE != 2) {
                  const float4 d4 = (MM3DGS_BWD2_PROBE & 8) ? make_float4(X0, Y0, X0, Y0) : ((const float4*)(wbase + LY::OFF_T))[pcol + yy * 4 + x];
                  c0 = __builtin_fmaf(uw.y, d4.x, c0);
                  c1 = __builtin_fmaf(uw.y, d4.y, c1);
                  c2 = __builtin_fmaf(uw.y, d4.z, c2);
                  if constexpr (MODE == 1) cz = __builtin_fmaf(uw.y, d4.w, cz);
                } else {
                  cz = __builtin_fmaf(uw.y, ((const float*)(wbase + LY::OFF_T))[pcol + yy * 4 + x], cz);
                }
                if constexpr (!Z45) cz5 = __builtin_fmaf(uw.y, t5[pcol + yy * 4 + x], cz5);
              }
              const float dyv = cxy.y - (Y0 + (float)(part * (SUB / 4) + yy));
              M0 = M0 + s0; Mx = Mx + s1; Mxx = Mxx + s2;
              My = __builtin_fmaf(s0, dyv, My);
              Mxy = __builtin_fmaf(s1, dyv, Mxy);
              Myy = __builtin_fmaf(s0 * dyv, dyv, Myy);
            }
            if constexpr (!Z45) cz = __builtin_fmaf(2.f * sCZ[r17 + ent].y, cz5, cz);      // d/dz of the [z, 1, z^2] bundle: sum w (dL3 + 2 z dL5)
            // merge the parts (ascending): afterwards every lane of the entry holds the totals
            auto merge = [&](float v) {
              if constexpr (NL == 2) return v + dpp_all<ROW_ROR8>(v);
              else { v = v + dpp_all<ROW_ROR4>(v); return v + dpp_all<ROW_ROR8>(v); }
            };
            M0 = merge(M0); Mx = merge(Mx); Mxx = merge(Mxx); My = merge(My); Mxy = merge(Mxy); Myy = merge(Myy);
            if constexpr (MODE != 0) cz = merge(cz);
            if constexpr (MODE != 2) { c0 = merge(c0); c1 = merge(c1); c2 = merge(c2); }
            if constexpr (POSE) {
              // dm = Kp (Mx, My) + Kq (Mxx, Mxy, Myy) + e_z cz;  the pose row collects dm (x) [x; 1]
              // (selects, not products: the lanes without an entry gathered record 0, which may never have been written)
              const float dm0 = on2 ? (k0.x * Mx + k0.y * My + k1.z * Mxx + k1.w * Mxy + k2.x * Myy) : 0.f;
              const float dm1 = on2 ? (k0.z * Mx + k0.w * My + k2.y * Mxx + k2.z * Mxy + k2.w * Myy) : 0.f;
              const float dm2 = on2 ? (k1.x * Mx + k1.y * My + k3.x * Mxx + k3.y * Mxy + k3.z * Myy + cz) : 0.f;
              const float x0 = on2 ? k3.w : 0.f, x1 = on2 ? k4.x : 0.f, x2 = on2 ? k4.y : 0.f;
              const bool owner = NL == 2 || (part & 1) == 0;          // four lanes per entry: parts 0 and 2 carry the two halves
              const bool hi = (NL == 2 ? part : (part >> 1)) != 0;
              const float ma = owner ? (hi ? dm2 : dm0) : 0.f, mb = owner ? (hi ? 1.f : dm1) : 0.f;
              // lanes 0-7 of the row (hi = 0): dR row 0 = dm0 x, dR row 1 = dm1 x;   lanes 8-15 (hi = 1): dR row 2 = dm2 x, dt = dm
              pacc[0] = __builtin_fmaf(ma, x0, pacc[0]); pacc[1] = __builtin_fmaf(ma, x1, pacc[1]); pacc[2] = __builtin_fmaf(ma, x2, pacc[2]);
              pacc[3] = __builtin_fmaf(mb, hi ? dm0 : x0, pacc[3]); pacc[4] = __builtin_fmaf(mb, hi ? dm1 : x1, pacc[4]); pacc[5] = __builtin_fmaf(mb, hi ? dm2 : x2, pacc[5]);
            } else
            if (on2 && !(MM3DGS_BWD2_PROBE & 2)) {
              float* const o = dsub + ((MM3DGS_BWD2_PROBE & 32) ? (size_t)lane : rbase + (size_t)(todo - 1u - (base + (uint32_t)ent))) * RECF;
              if constexpr (MODE == 0) {      // [Mx My Mxx Mxy | Myy sum(G dL/dalpha) c0 c1 | c2], the moments scaled by the splat's opacity (preprocess_bwd's layout)
                const float op = sB[r17 + ent].y;
                if (part == 0) { const f4u v0 = {op * Mx, op * My, op * Mxx, op * Mxy}; *(f4u*)o = v0; o[8] = c2; }
                if (part == 1) { const f4u v1 = {op * Myy, M0, c0, c1}; *(f4u*)(o + 4) = v1; }
              } else
              if constexpr (MODE == 1) {      // [M0 Mx Mxx c0 | c1 c2 cz My | Mxy Myy]
                if (MM3DGS_BWD2_PROBE & 16) { if (part == 0) { const f4u v0 = {M0 + Mx + Mxx + c0, c1 + c2 + cz + My, Mxy, Myy}; *(f4u*)o = v0; } } else {
                if (part == 0) { const f4u v0 = {M0, Mx, Mxx, c0}; *(f4u*)o = v0; }
                if (part == 1) { const f4u v1 = {c1, c2, cz, My}; *(f4u*)(o + 4) = v1; }
                if (part == (NL == 2 ? 0 : 2)) { const f2u v2 = {Mxy, Myy}; *(f2u*)(o + 8) = v2; } }
              } else {                        // [M0 Mx Mxx cz | My Mxy Myy]
                if (part == 0) { const f4u v0 = {M0, Mx, Mxx, cz}; *(f4u*)o = v0; }
                if (part == 1) { const f2u v1 = {My, Mxy}; *(f2u*)(o + 4) = v1; o[6] = Myy; }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();      // the tile is rewritten by the next sub-chunk's phase 1
        }
        if (base + CH >= maxtodo) break;      // (wave-uniform)
        sA[slane] = rec_n.A; sB[slane] = rec_n.B; sCZ[slane] = make_float2(rec_n.C.x, rec_n.C.y);
        my_idx = POSE ? ent_nxt.x : ent_nxt.y;
        ent_nxt = ent_nn;
      }
    };
    wave_prio_by_steps(maxtodo, mean_steps_b);
    if (z45_wave) run2(std::true_type{});
    else run2(std::false_type{});
    wave_prio_reset();
    if (cam.stats && lane == 0) { atomicAdd(&iv.hdr->bwd_wave_visits, n_visit); atomicAdd(&iv.hdr->bwd_wave_iters, n_visit); }
  }
  } else
  if (maxtodo != 0 && !PROBE(cam, 9)) {   // wave-uniform; (probe builds, bit 9: timing without the main loop) (a wave without work still takes part in the workgroup's per-tile combine below)

  const int my_slot = MODE == 0 ? WaveReduce<NV>::slot(q) : (SEP_REDUCE2 ? SepReduce2<MODE == 1>::slot(q) : SepReduce<MODE == 1>::slot(q));
  float ym_0 = 0.f, ym_1a = 0.f, ym_1b = 0.f;
  if (MODE != 0 && !SEP_REDUCE2) SepReduce<MODE == 1>::ymult(q, ym_0, ym_1a, ym_1b);
  // this lane's component of record 0; the list entries carry the record index of their (splat, block)
  float* const my_rec = dsub + (my_slot >= 0 ? my_slot : 0);
  uint32_t n_visit = 0, n_red = 0;

  // chunk c of a row holds its list entries todo-1-(16c+q): entry order == traversal order (back to front)
  {
    const uint2 e0 = (uint32_t)q < todo ? list[todo - 1u - q] : make_uint2(0u, 0u);
    SplatRec r0 = load_rec<C>(g.splat, e0.x, (uint32_t)q < todo);
    // SLAM modes: the entry's gradient-record index rides in the staged record's constant field (C.z = the "1" of [z, 1, z^2]): one LDS
    // read and its address less per (row, splat) step than a separate index array
    if constexpr (MODE != 0) r0.C.z = __uint_as_float(e0.y);
    stg[0][wv][0][slane] = r0.A;
    stg[0][wv][1][slane] = r0.B;
    if (C > 2) stg[0][wv][2][slane] = r0.C;
  }
  uint2 ent_nxt = CH + q < todo ? list[todo - 1u - (CH + q)] : make_uint2(0u, 0u);
  int cur = 0;

  // The SLAM losses leave the silhouette and depth^2 channels without gradient (dL[4] = dL[5] = 0): a wave that sees only
  // zeros there runs a loop instance with those terms removed (exact: they would multiply by zero).
  // Likewise the background term of dL/dalpha, -T_final (bg . dL) / (1 - alpha): over a black background (the shipped configurations) it is
  // zero for every pixel, and the same instance drops it (exact: x - 0 * r = x for the finite r = 1 / (1 - alpha), alpha <= 0.99).
  const bool z45_wave = MODE != 0 && __ballot(dL[C > 4 ? 4 : 0] != 0.f || dL[C > 5 ? 5 : 0] != 0.f || Tf_bg != 0.f) == 0ull;
  // A splat counts for this lane's pixel while  pos = todo - 1 - step < last_contributor,  step = base + j the wave-uniform position in the
  // traversal:  step >= first_step  with the per-lane constant below -- one compare against a scalar instead of a subtraction and a compare per
  // step.  (todo >= last_contributor unless the list was clamped to `count`; then first_step = 0: every listed splat counts.)
  const uint32_t first_step = todo - min(todo, last_contributor);
  const uint32_t mean_steps_b = wave_mean_steps(cam, iv);
  auto run_chunks = [&](auto z45_tag) {
  constexpr bool Z45 = decltype(z45_tag)::value && MODE != 0;   // (SLAM modes have C == 6)
  for (uint32_t base = 0; base < maxtodo; base += CH, cur ^= 1) {
    SplatRec rec_n = load_rec<C>(g.splat, ent_nxt.x, base + CH + q < todo);
    if constexpr (MODE != 0) rec_n.C.z = __uint_as_float(ent_nxt.y);
    const uint2 ent_nn = base + 2 * CH + q < todo ? list[todo - 1u - (base + 2 * CH + q)] : make_uint2(0u, 0u);
    const float4 (*wS)[STG_N] = stg[cur][wv];
    const int r16 = row * STG_ROW;
    __builtin_amdgcn_wave_barrier();
    const int cnt = __builtin_amdgcn_readfirstlane((int)min(CH, maxtodo - base));
    auto splat_bwd = [&](const float4& A, const float4& B, const float4& Cc, const int j) {
      const uint32_t step = base + (uint32_t)j;               // wave-uniform
      const size_t ti = rbase + (size_t)(todo - 1u - step);      // list-major records (every mode since round 6)
      const bool row_on = step < todo;                        // this row still has an entry at this step
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float power = splat_power(dx, dy, A.z, A.w, B.x);
      const float G = SPLAT_EXP(power);
      const float araw = B.y * G;
      const float alpha = fminf(0.99f, araw);
      const bool valid = row_on && (step >= first_step) && !(power > 0.f) && !(alpha < ALPHA_MIN);
      float tot = 0.f;
      n_visit++;
      if (MODE != 0 || __ballot(valid) != 0ull) {   // SLAM modes: four rows with different splats -- a whole-wave miss is rare, the vote is not worth its cost
        n_red++;
        // two selects (a_eff, G_eff).  ONE select of the un-clamped o G with alpha = min(0.99, .) of it is two instructions and six registers
        // less and measured 1.3 us SLOWER: the select moves into the dependent chain alpha -> 1 - alpha -> rcp -> T
        const float a_eff = valid ? alpha : 0.f;
        const float G_eff = valid ? G : 0.f;
        // generic mode: the reciprocal to within an ulp -- T is rebuilt by ~50 successive divisions per pixel and the raw 1-ulp v_rcp_f32 showed up as
        // 5e-6 of noise on every gradient (camera gradients are held to 1e-5).  v_rcp_f32 + one Newton step holds the same bars as the IEEE
        // division sequence (tests/test_gpu_parity.py) at 3 instead of ~10 instructions: 1080p / 3 M backward compositor 1327 -> 1291 us.
        // The SLAM modes keep the raw v_rcp_f32 (an exact division changes none of their parity figures: measured, round 3).
        float r;
        if constexpr (MODE == 0) {
          const float d = 1.f - a_eff, r0 = __builtin_amdgcn_rcpf(d);
          r = fmaf(fmaf(-d, r0, 1.f), r0, r0);
        } else {
          r = __builtin_amdgcn_rcpf(1.f - a_eff);
        }
        Tr *= r;  // transmittance in front of this splat
        const float w = a_eff * Tr;
        float col[C];
        if constexpr (C > 0) col[0] = B.z;
        if constexpr (C > 1) col[1] = B.w;
        if constexpr (C > 2) col[2] = Cc.x;
        if constexpr (C > 3) col[3] = Cc.y;
        if constexpr (C > 4) col[4] = MODE == 0 ? Cc.z : 1.f;      // (SLAM modes: that field carries the record index; the channel is the constant 1)
        if constexpr (C > 5) col[5] = Cc.w;
        // dL/dalpha needs sum_ch (c_ch - behind_ch) dL_ch: track the dL-weighted colour behind as ONE scalar
        // (behind_dot) instead of C running colours: qd = c . dL;  dLa = qd - behind_dot;  behind_dot += a (qd - behind_dot)
        float qd = 0.f;
#pragma unroll
        for (int ch = 0; ch < (Z45 ? 4 : C); ch++) qd = fmaf(col[ch], dL[ch], qd);
        const float diff = qd - behind_dot;
        behind_dot = fmaf(a_eff, diff, behind_dot);
        const float dLa = Z45 ? diff * Tr : diff * Tr - Tf_bg * r;
        // screen-space geometry: only the moments of u = dL/dG * G are reduced; the consumer (preprocess_bwd) turns them
        // into d/dxy and d/dconic with the splat's own conic:  dxy = -(Q m1),  dconic = -(1/2 m_xx, m_xy, 1/2 m_yy)
        const float u = B.y * dLa * G_eff;
        if constexpr (MODE == 0) {
          float vals[NV];
#pragma unroll
          for (int ch = 0; ch < C; ch++) vals[6 + ch] = w * dL[ch];
          const float mx = u * dx, my = u * dy;
          vals[0] = mx;
          vals[1] = my;
          vals[2] = mx * dx;
          vals[3] = mx * dy;
          vals[4] = my * dy;
          vals[5] = G_eff * dLa;
          tot = WaveReduce<NV>::run(vals, lane);
        } else {
          // SLAM records: the zeroth moment M0 = sum u also carries the opacity gradient (sum G dL/dalpha = M0 / opacity)
          const float cz = Z45 ? w * dL[3] : w * fmaf(2.f * col[3], dL[5], dL[3]);   // d/dz of the [z, 1, z^2] bundle, chained here
          if constexpr (SEP_REDUCE2) tot = SepReduce2<MODE == 1>::run(u, dx, dy, w * dL[0], w * dL[1], w * dL[2], cz, lane);
          else tot = SepReduce<MODE == 1>::run(u, dx, dy, w * dL[0], w * dL[1], w * dL[2], cz, lane, ym_0, ym_1a, ym_1b);
        }
      }
      // this row is the only writer of the (block, splat) record: up to 12 of its lanes store 48 contiguous bytes
      if (my_slot >= 0 && row_on && !PROBE(cam, 0)) my_rec[ti * RECF] = tot;
    };
    // two register sets used alternately: the next splat's LDS reads are in flight while the current one is evaluated,
    // without any register-to-register copies
    float4 A0 = wS[0][r16], B0 = wS[1][r16], C0 = wS[2][r16];
    for (int j = 0; j < cnt; j += 2) {
      const int j1 = j + 1 < cnt ? j + 1 : j;
      const float4 A1 = wS[0][r16 + j1], B1 = wS[1][r16 + j1], C1 = wS[2][r16 + j1];
      splat_bwd(A0, B0, C0, j);
      if (j + 1 < cnt) {
        const int j2 = j + 2 < cnt ? j + 2 : j1;
        A0 = wS[0][r16 + j2]; B0 = wS[1][r16 + j2]; C0 = wS[2][r16 + j2];
        splat_bwd(A1, B1, C1, j1);
      }
    }
    stg[cur ^ 1][wv][0][slane] = rec_n.A;
    stg[cur ^ 1][wv][1][slane] = rec_n.B;
    if (C > 2) stg[cur ^ 1][wv][2][slane] = rec_n.C;
    ent_nxt = ent_nn;
  }
  };
  wave_prio_by_steps(maxtodo, mean_steps_b);
  if (z45_wave) run_chunks(std::true_type{});
  else run_chunks(std::false_type{});
  wave_prio_reset();      // (the combine at top priority instead: no gain, 56.8 -> 57.0 us)
  if (cam.stats && lane == 0) {
    atomicAdd(&iv.hdr->bwd_wave_visits, n_visit);
    atomicAdd(&iv.hdr->bwd_wave_iters, n_red);
  }
  }   // maxtodo != 0
  if constexpr (POSE) {
    // ---- the tile's pose row: lanes 0-7 of every row hold their share of components 0-5, lanes 8-15 of components 6-11 (zeros in the lanes that
    // carried nothing); double from here on -- the terms cancel across a tile -- in a fixed order: inside the 8-lane groups, across the four rows,
    // across the four waves
    double dsum[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double v = (double)pacc[k];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      dsum[k] = v;
    }
    __syncthreads();      // every wave has left its main loop: the staging memory is free
    double* const wtot = (double*)smem_raw;      // [4 waves][12]
    if (lane == 0 || lane == 8) {
#pragma unroll
      for (int k = 0; k < 6; k++) wtot[wv * 12 + (lane == 8 ? 6 : 0) + k] = dsum[k];
    }
    __syncthreads();
    if (tid < 12) dsub[(size_t)tile * 32 + tid] = (float)((wtot[tid] + wtot[12 + tid]) + (wtot[24 + tid] + wtot[36 + tid]));
  }
  if constexpr (!POSE) {
    // ---- per-tile combine (every mode since round 6; generic mode: records of 6 + C floats): one record per (tile, splat) pair = the sum of the pair's block records, in ascending
    // block order (deterministic).  The backward projection then reads ONE record per pair (contiguous per Gaussian) instead of one
    // per listed 4x4 block: a quarter of the bytes on the kernel that the counters show to be bandwidth bound on exactly them
    // (148 MB per mapping launch, 56 MB of it block records).  The block records were written by this workgroup's own waves a
    // moment ago (same CU: visible after the barrier); a pair is found by its position in the tile's bin (payload / trec).
    // The lane's first table entry is requested BEFORE the barrier (round 5): it depends on nothing the rows compute, the main loop's registers
    // are dead here, and the load lands while the wave waits for the tile's slowest wave -- one of the combine's three dependent round trips
    // (table -> records -> store) off the tail every workgroup ends with (the skeleton probes: 12 of the mapping launch's 57 us are this pass).
    // (Measured and rejected in the same round: the pairs whose listed blocks all lie in ONE 8x8 sub-tile summed by that sub-tile's wave BEFORE
    // the barrier -- every wave then scans the whole bin for its own pairs behind an s_waitcnt vmcnt(0): mapping launch 59.4 us against 58.1,
    // fused tracking kernel 71.3 against 67.1, profiles/r05_ab_combine.txt.)
    float* __restrict__ dtile = dsub + (size_t)NLIST * (size_t)N_cap * SPLAT_F;
    // (Also measured and rejected in round 5: four lanes per pair, one per 16-byte chunk of a record, so that one load instruction fetches a record
    // with adjacent lanes -- a third of the cache-line requests, but a lane group then walks four pairs one after the other: mapping launch 58.6 us
    // against 58.1, fused tracking kernel 69.2 against 67.1.  The pass is a latency chain per pair, not a request-rate limit: one pair per lane it is.)
    // Round 6: the block records are LIST-major (record of the k-th entry of block list L = NLIST * start + L * len + k), so a pair's records are found
    // through its list positions -- the rank of the pair among the bin's entries that list block L, in SORTED order.  The sort left the bin in sorted
    // order over the keys (block mask | per-tile record << 32); it is walked in 64-entry chunks (chunk c: wave c % 4), the sixteen running list lengths
    // come from ballots, and a wave sums the records of the chunks it owns, each lane one pair, the pair's blocks in ascending order, four records in
    // flight -- the order of the sums is what it was.
    const unsigned long long* __restrict__ sorted = b.keys + start;
    // Round 6, second half: the scan is DISTRIBUTED.  Chunk c belongs to wave c % 4.  Before the barrier (phase A) a wave counts, for each of ITS chunks,
    // how many entries list each of the sixteen blocks, and leaves the counts in its own slice of the staging memory (free once its main loop is done);
    // after it (phase B) lanes 0-15 of every wave run over all chunks' counts -- a few dozen LDS bytes -- and the wave recomputes the ballots of its
    // own chunks only.  Every wave used to ballot through ALL chunks for the running list lengths: 2 300 vector instructions per wave on a 1080p tile
    // of a 3 M-Gaussian map (18 chunks), a fifth of the pass there (profiles/r06_c5_probes.txt).  Tiles beyond COMB_DIST_CHUNKS chunks keep that form.
    // The words of a wave's first WPRE chunks are requested before the barrier (they depend on nothing the rows compute, the main loop's registers are
    // dead): a tile of up to 256 WPRE pairs -- every tile of a SLAM map -- needs no load after it.
    constexpr int WPRE = (MODE == 0 && C > 4) ? 4 : 8;      // (the 11 / 12-float generic records leave registers for four)
    constexpr uint32_t COMB_OWN_MAX = 16;                    // own chunks whose counts fit the wave's scratch: tiles of up to 4096 pairs
    constexpr uint32_t COMB_DIST_CHUNKS = 4 * COMB_OWN_MAX;
    constexpr size_t WSLICE = BWD_IS_TWO_PHASE(MODE, C) ? (size_t)Bwd2Lds<(MODE == 0 ? 1 : MODE)>::SLICE : sizeof(float4) * 3 * STG_N;   // a wave's private piece of the staging memory
    static_assert(WSLICE >= COMB_OWN_MAX * NLIST + 64 * (NLIST / 2) * 4, "a wave's slice holds its chunk counts and its list positions");
    const uint32_t nchunks = (len + 63u) >> 6;
    const bool dist = nchunks <= COMB_DIST_CHUNKS;          // workgroup-uniform
    unsigned char* const cnt_mine = smem_raw + (size_t)wv * WSLICE;      // u8[own chunk][list]
    unsigned long long wpre[WPRE];
#pragma unroll
    for (int k = 0; k < WPRE; k++) wpre[k] = 0ull;
    auto own_word = [&](uint32_t k, uint32_t i) -> unsigned long long {      // word of this wave's k-th chunk (entry i of the bin)
      unsigned long long wd = 0ull;
#pragma unroll
      for (int t = 0; t < WPRE; t++) wd = k == (uint32_t)t ? wpre[t] : wd;
      if (k >= (uint32_t)WPRE) wd = i < len ? sorted[i] : 0ull;
      return wd;
    };
    if (!PROBE(cam, 5)) {
#pragma unroll
      for (int k = 0; k < WPRE; k++) {
        const uint32_t i = (uint32_t)((4 * k + wv) * 64 + lane);
        if (i < len) wpre[k] = sorted[i];
      }
      if (dist) {      // phase A
        for (uint32_t k = 0, c = (uint32_t)wv; c < nchunks; k++, c += 4u) {
          const uint32_t i = c * 64u + (uint32_t)lane;
          const unsigned long long wd = own_word(k, i);
          const uint32_t mask = i < len ? ((uint32_t)wd & 0xffffu) : 0u;
          unsigned long long mine = 0ull;
#pragma unroll
          for (int Lq = 0; Lq < NLIST; Lq++) {
            const unsigned long long bal = __ballot((mask >> Lq) & 1u);
            mine = lane == Lq ? bal : mine;
          }
          if (lane < NLIST) cnt_mine[k * NLIST + (uint32_t)lane] = (unsigned char)__popcll(mine);
        }
      }
    }
    __syncthreads();
    if (!PROBE(cam, 5)) {      // (probe builds, bit 5: timing without the combine)
      // wave-private scratch behind the counts: the lane's sixteen list positions of the chunk, as u16
      uint32_t* const pos16 = (uint32_t*)(cnt_mine + COMB_OWN_MAX * NLIST) + lane * (NLIST / 2);      // (two positions per word)
      const unsigned long long lt = (1ull << lane) - 1ull;
      const bool wide = len > 0xffffu;      // (list positions fit 16 bits: a direct-bin span holds at most 8191 pairs; longer packed-bin lists take the slow path)
      // sum of the pair's block records (its blocks in ascending order, four records in flight) and the store of its per-tile record
      auto finish_chunk = [&](const uint32_t i, uint32_t mask, const uint32_t tr, const uint32_t (&pos)[NLIST]) {
        if (!wide && MODE != 0) {
#pragma unroll
          for (int Lq = 0; Lq < NLIST; Lq += 2) pos16[Lq >> 1] = pos[Lq] | (pos[Lq + 1] << 16);
        }
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
        if constexpr (MODE == 0) {
          // generic mode (long lists: a 1080p tile of a 3 M-Gaussian map holds ~1100 pairs, ~4500 block records): the lanes visit the sixteen lists in
          // LOCKSTEP, four at a time.  The entries of one chunk that list block L hold CONSECUTIVE positions of list L, so the lanes that read in a step
          // read one contiguous span of records and the load unit merges them into a few cache-line requests; with every lane following its own lowest
          // set bit (the SLAM form below: one round trip for most pairs) each 16-byte piece of each record is a request of its own, and at this size the
          // pass is bound by the L2's request rate: 111 M of them.  Same summation order (ascending block), same sums.  (The SLAM modes lose with this form, also
          // on the 775-pair tiles of configs[3]: mapping backward 221 - 233 -> 248 - 266 us -- four dependent round trips per chunk instead of one or two.)
          // (Also measured and rejected: the idle lanes of a step reading a record of ZEROS and adding it like the others -- twelve selects per record slot less, but
          //  two possible base addresses turn the loads' scalar-base addressing into 64-bit vector arithmetic: generic launch + 20 us, SLAM mapping launch + 1.5.)
#pragma unroll
          for (int g4 = 0; g4 < NLIST; g4 += 4) {
            if (__ballot((mask >> g4) & 0xfu) == 0ull) continue;
            float4 ra[4], rb[4], rc[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int Lq = g4 + u;
              on[u] = (mask >> Lq) & 1u;
              const size_t rec = (size_t)NLIST * start + (size_t)Lq * len + (size_t)pos[Lq];
              const float* r = dsub + ((on[u] && !PROBE(cam, 4)) ? rec * RECF : (size_t)0);
              ra[u] = ld4u(r); rb[u] = ld4u(r + 4);
              rc[u] = NV > 8 ? ld4u(r + 8) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              a0.x += on[u] ? ra[u].x : 0.f; a0.y += on[u] ? ra[u].y : 0.f; a0.z += on[u] ? ra[u].z : 0.f; a0.w += on[u] ? ra[u].w : 0.f;
              a1.x += on[u] ? rb[u].x : 0.f; a1.y += on[u] ? rb[u].y : 0.f; a1.z += on[u] ? rb[u].z : 0.f; a1.w += on[u] ? rb[u].w : 0.f;
              if (NV > 8) a2.x += on[u] ? rc[u].x : 0.f;
              if (NV > 9) a2.y += on[u] ? rc[u].y : 0.f;
              if (NV > 10) a2.z += on[u] ? rc[u].z : 0.f;
              if (NV > 11) a2.w += on[u] ? rc[u].w : 0.f;
            }
          }
          mask = 0u;
        }
        while (__ballot(mask != 0u) != 0ull) {      // a pair lists ~4 blocks on average: one round for most
          constexpr int UR = 4;
          float4 ra[UR], rb[UR], rc[UR];
          bool on[UR];
#pragma unroll
          for (int u = 0; u < UR; u++) {
            on[u] = mask != 0u;
            const int Lq = on[u] ? __ffs((int)mask) - 1 : 0;
            mask &= mask - 1u;
            uint32_t pq;
            if (!wide) { const uint32_t w2 = pos16[Lq >> 1]; pq = (Lq & 1) ? (w2 >> 16) : (w2 & 0xffffu); }
            else { pq = 0u;
#pragma unroll
              for (int t = 0; t < NLIST; t++) pq = Lq == t ? pos[t] : pq; }
            const size_t rec = (size_t)NLIST * start + (size_t)Lq * len + (size_t)pq;
            const float* r = dsub + ((on[u] && !PROBE(cam, 4)) ? rec * RECF : (size_t)0);      // (probe builds, bit 4: timing without the record gather)
            // (a record shorter than the twelve floats read: the rest belongs to the next record -- or, behind the last one, to the per-tile region --
            //  and is never stored)
            ra[u] = ld4u(r); rb[u] = ld4u(r + 4);
            rc[u] = NV > 8 ? ld4u(r + 8) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < UR; u++) {
            a0.x += on[u] ? ra[u].x : 0.f; a0.y += on[u] ? ra[u].y : 0.f; a0.z += on[u] ? ra[u].z : 0.f; a0.w += on[u] ? ra[u].w : 0.f;
            a1.x += on[u] ? rb[u].x : 0.f; a1.y += on[u] ? rb[u].y : 0.f; a1.z += on[u] ? rb[u].z : 0.f; a1.w += on[u] ? rb[u].w : 0.f;
            if (NV > 8) a2.x += on[u] ? rc[u].x : 0.f;
            if (NV > 9) a2.y += on[u] ? rc[u].y : 0.f;
            if (NV > 10) a2.z += on[u] ? rc[u].z : 0.f;
            if (NV > 11) a2.w += on[u] ? rc[u].w : 0.f;
          }
        }
        if (i < len && tr != 0xffffffffu && !PROBE(cam, 11)) {      // (probe builds, bit 11: combine without its stores)
          // exactly NV floats, in the widest pieces (mapping: 4 + 4 + 2, tracking: 4 + 2 + 1, generic: 4 + ...): the next record is another lane's
          float* o = dtile + (size_t)tr * RECF;
          const float v[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
          const f4u q0 = {v[0], v[1], v[2], v[3]};
          *(f4u*)o = q0;
          constexpr int F1 = NV >= 8 ? 8 : 4;      // floats stored after the second piece
          if constexpr (NV >= 8) { const f4u q1 = {v[4], v[5], v[6], v[7]}; *(f4u*)(o + 4) = q1; }
          constexpr int F2 = NV - F1 >= 4 ? F1 + 4 : (NV - F1 >= 2 ? F1 + 2 : F1);
          if constexpr (NV - F1 >= 4) { const f4u q2 = {v[F1], v[F1 + 1], v[F1 + 2], v[F1 + 3]}; *(f4u*)(o + F1) = q2; }
          else if constexpr (NV - F1 >= 2) { const f2u q2 = {v[F1], v[F1 + 1]}; *(f2u*)(o + F1) = q2; }
          constexpr int F3 = NV - F2 >= 2 ? F2 + 2 : F2;
          if constexpr (NV - F2 >= 2) { const f2u q3 = {v[F2], v[F2 + 1]}; *(f2u*)(o + F2) = q3; }
          if constexpr (NV - F3 >= 1) o[F3] = v[F3];
        }
      };
      if (dist) {
        // phase B: lane L < 16 carries the running length of list L over the chunks in order; at one of its own chunks the wave takes the sixteen
        // lengths out of those lanes and adds each entry's rank inside the chunk
        uint32_t acc = 0u;
        for (uint32_t c = 0; c < nchunks; c++) {
          const uint32_t w = c & 3u, k = c >> 2;
          const uint32_t cv = lane < NLIST ? (uint32_t)smem_raw[(size_t)w * WSLICE + k * NLIST + (uint32_t)lane] : 0u;
          if (w == (uint32_t)wv) {      // wave-uniform
            const uint32_t i = c * 64u + (uint32_t)lane;
            const unsigned long long wd = own_word(k, i);
            const uint32_t mask = i < len ? ((uint32_t)wd & 0xffffu) : 0u;
            uint32_t pos[NLIST];
#pragma unroll
            for (int Lq = 0; Lq < NLIST; Lq++) {
              const unsigned long long bal = __ballot((mask >> Lq) & 1u);
              pos[Lq] = (uint32_t)__builtin_amdgcn_readlane((int)acc, Lq) + (uint32_t)__popcll(bal & lt);
            }
            finish_chunk(i, mask, (uint32_t)(wd >> 32), pos);
          }
          acc += cv;
        }
      } else {
        // (a tile of more than 4096 pairs: every wave ballots through all chunks for the running list lengths)
        uint32_t run[NLIST];
#pragma unroll
        for (int Lq = 0; Lq < NLIST; Lq++) run[Lq] = 0u;
        for (uint32_t c = 0; c < nchunks; c++) {
          const uint32_t i = c * 64u + (uint32_t)lane;
          const unsigned long long wd = i < len ? sorted[i] : 0ull;
          const uint32_t mask = i < len ? ((uint32_t)wd & 0xffffu) : 0u;
          uint32_t pos[NLIST];
#pragma unroll
          for (int Lq = 0; Lq < NLIST; Lq++) {
            const unsigned long long bal = __ballot((mask >> Lq) & 1u);
            pos[Lq] = run[Lq] + (uint32_t)__popcll(bal & lt);
            run[Lq] += (uint32_t)__popcll(bal);
          }
          if ((c & 3u) == (uint32_t)wv) finish_chunk(i, mask, (uint32_t)(wd >> 32), pos);      // wave-uniform
        }
      }
    }
  }
}

template <int C, int MODE, bool POSE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
composite_bwd_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, const float* __restrict__ dL_dout,
                     float* __restrict__ dsub, int has_tl, TrackLoss tl, int dl_planes) {
  constexpr size_t LOSS_BYTES = MODE == 1 ? sizeof(LossGradSmem) + 4 * 256 * sizeof(float) : 0;
  // (generic mode: 32 KB on purpose -- five workgroups per CU; with the 26 KB the staging buffers need, six fit and the 1080p /
  //  3 M-Gaussian pass ran 14 % slower: every workgroup gathers ~1000 splat records by id, and six of them overflow the L1)
  constexpr size_t MIN_BYTES = MODE == 0 ? 32768 : 0;
  constexpr size_t STG_BYTES = BWD_IS_TWO_PHASE(MODE, C) ? (size_t)BWD2_BYTES(MODE == 0 ? 1 : MODE) : (size_t)BWD_STG_BYTES;
  constexpr size_t NEED = STG_BYTES > LOSS_BYTES ? STG_BYTES : LOSS_BYTES;
  __shared__ __align__(16) unsigned char smem_raw[NEED > MIN_BYTES ? NEED : MIN_BYTES];
  const int T = cam.gx * cam.gy;
  const int tile = slam_tile(cam, iv, blockIdx.x, T);
  if (tile >= T) return;
  // every launch of this iteration's forward has retired: what the sticky overflow word holds NOW is what the whole backward projection that
  // follows must act on (Mm3dgsHeader.overflow_seen; its fused second half bins the next view and may raise the word itself)
  if (MODE == 1 && blockIdx.x == 0 && threadIdx.x == 0) iv.hdr->overflow_seen = iv.hdr->overflow;
  composite_bwd_body<C, MODE, POSE>(tile, cam, g, iv, b, N_cap, dL_dout, dsub, has_tl, tl, dl_planes, smem_raw);
}

// A tracking iteration with the masked-L1 loss alone (its 1/n is applied to the pose gradient afterwards, so the per-pixel loss
// gradient needs nothing from other tiles): sort, forward compositing and backward compositing of a tile in ONE launch.  The
// backward pass picks its pixel's final transmittance, contributor count and colours up from memory the same lane wrote a
// moment ago; one launch, its ramp and the backward prologue's cold misses less per iteration.
template <bool POSE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
sort_composite_fwd_bwd_track_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, float* __restrict__ out, int clean,
                                    TrackLoss tl, int direct_blocks, float* __restrict__ dsub, uint32_t direct_cap, int slot_bits) {
  constexpr size_t TRACK_BWD_BYTES = BWD_TWO_PHASE ? (size_t)BWD2_BYTES(2) : (size_t)BWD_STG_BYTES;
  __shared__ __align__(16) unsigned char smem[TRACK_BWD_BYTES];     // >= forward staging (25.5 KB) >= sort keys + runs + payloads (24 KB)
  __shared__ SortShared sh;
  __shared__ double red[4][12];
  static_assert(TRACK_BWD_BYTES >= sizeof(float4) * 2 * 4 * 3 * STG_N, "LDS union too small for the forward staging buffers");
  static_assert(TRACK_BWD_BYTES >= 3 * RANK_SORT_MAX * sizeof(unsigned long long) + sizeof(SortEmit), "LDS union too small for the sort's keys, runs, payloads and the emission's counters");
  const int T = cam.gx * cam.gy;
  const int tile = slam_tile(cam, iv, blockIdx.x, T);
  if (tile >= T) return;
  sort_tile_body<2048, true>(tile, cam.gx, 0, g, iv, b, N_cap, clean, (unsigned long long*)smem, sh, 0, direct_blocks, direct_cap, slot_bits);
  __syncthreads();
  composite_fwd_body<6>(tile, cam, g, iv, b, N_cap, out, (float4 (*)[4][3][STG_N])smem, sh.run, &tl, red, &sh);
  __syncthreads();   // out / final_T / n_contrib of the tile are written, the staging memory is free
  composite_bwd_body<6, 2, POSE>(tile, cam, g, iv, b, N_cap, nullptr, dsub, 1, tl, 6, smem, &sh);
}

template <int C>
static void launch_fwd_c(const CamDev& cam, GeomView g, ImageView iv, BinView b, uint32_t ncap, float* out, hipStream_t s) {
  int T = cam.gx * cam.gy;
  int grid = ((T + 7) / 8) * 8;
  static const int pad = env_flag("MM3DGS_FWD_LDS_PAD", 0);
  hipLaunchKernelGGL((composite_fwd_kernel<C>), dim3(grid), dim3(256), (size_t)pad, s, cam, g, iv, b, ncap, out);
}
template <int C>
static void launch_bwd_c(const CamDev& cam, GeomView g, ImageView iv, BinView b, uint32_t ncap, const float* dL, float* dsub,
                         hipStream_t s) {
  int T = cam.gx * cam.gy;
  int grid = ((T + 7) / 8) * 8;
  TrackLoss none = {};
  // (dynamic LDS is never touched: it only limits how many workgroups share a CU, see composite_bwd_kernel)
  static const int pad = env_flag("MM3DGS_BWD_LDS_PAD", 0);
  hipLaunchKernelGGL((composite_bwd_kernel<C, 0>), dim3(grid), dim3(256), (size_t)pad, s, cam, g, iv, b, ncap, dL, dsub, 0, none, C);
}
// MM3DGS_SLAM_LDS_PAD (developer experiment): bytes of never-touched dynamic LDS added to the SLAM compositor launches -- it only lowers
// how many workgroups share a CU (occupancy sensitivity of the kernels; results are unaffected)
static size_t slam_lds_pad() { static const int pad = env_flag("MM3DGS_SLAM_LDS_PAD", 0); return (size_t)pad; }
void launch_composite_bwd_slam(const CamDev& cam, bool tracking, GeomView g, ImageView iv, BinView b, size_t N_cap, const float* dL,
                               float* dsub, hipStream_t s, const TrackLoss* tl, int dl_planes, bool pose_chain) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  int T = cam.gx * cam.gy;
  int grid = slam_grid(cam, T);
  TrackLoss none = {};
#if BWD_TWO_PHASE
  if (tracking && pose_chain)
    hipLaunchKernelGGL((composite_bwd_kernel<6, 2, true>), dim3(grid), dim3(256), slam_lds_pad(), s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
  else
#endif
  if (tracking)
    hipLaunchKernelGGL((composite_bwd_kernel<6, 2>), dim3(grid), dim3(256), slam_lds_pad(), s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
  else
    hipLaunchKernelGGL((composite_bwd_kernel<6, 1>), dim3(grid), dim3(256), slam_lds_pad(), s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
}

void launch_sort_composite_fwd6(const CamDev& cam, GeomView g, ImageView iv, BinView b, size_t N_cap, float* out, int clean, hipStream_t s,
                                const TrackLoss* tl, int direct_blocks, uint32_t direct_cap, int slot_bits) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  int T = cam.gx * cam.gy;
  int grid = slam_grid(cam, T);
  TrackLoss none = {};
  hipLaunchKernelGGL((sort_composite_fwd_kernel<6>), dim3(grid), dim3(256), slam_lds_pad(), s, cam, g, iv, b, ncap, out, clean, tl ? 1 : 0, tl ? *tl : none, direct_blocks, direct_cap, slot_bits);
}

void launch_sort_composite_fwd_bwd_track(const CamDev& cam, GeomView g, ImageView iv, BinView b, size_t N_cap, float* out, int clean,
                                         hipStream_t s, const TrackLoss& tl, int direct_blocks, float* dsub, uint32_t direct_cap, int slot_bits, bool pose_chain) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  int T = cam.gx * cam.gy;
  int grid = slam_grid(cam, T);
#if BWD_TWO_PHASE
  if (pose_chain)
    hipLaunchKernelGGL(sort_composite_fwd_bwd_track_kernel<true>, dim3(grid), dim3(256), slam_lds_pad(), s, cam, g, iv, b, ncap, out, clean, tl, direct_blocks, dsub, direct_cap, slot_bits);
  else
#endif
  hipLaunchKernelGGL(sort_composite_fwd_bwd_track_kernel<false>, dim3(grid), dim3(256), slam_lds_pad(), s, cam, g, iv, b, ncap, out, clean, tl, direct_blocks, dsub, direct_cap, slot_bits);
}
// whether this build's tracking compositors carry the pose chain (the one-phase A/B build does not)
bool composite_has_pose_chain() { return BWD_TWO_PHASE != 0; }

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
                          float* dsub, hipStream_t s) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  switch (C) {
    case 1: launch_bwd_c<1>(cam, g, iv, b, ncap, dL, dsub, s); break;
    case 2: launch_bwd_c<2>(cam, g, iv, b, ncap, dL, dsub, s); break;
    case 3: launch_bwd_c<3>(cam, g, iv, b, ncap, dL, dsub, s); break;
    case 4: launch_bwd_c<4>(cam, g, iv, b, ncap, dL, dsub, s); break;
    case 5: launch_bwd_c<5>(cam, g, iv, b, ncap, dL, dsub, s); break;
    default: launch_bwd_c<6>(cam, g, iv, b, ncap, dL, dsub, s); break;
  }
}
