// Backward compositor of the SLAM modes, second generation: two pixels per lane, eight 4x4 blocks per wave, and the per-block
// reduction of the gradient terms on the matrix cores.
//
// The first-generation kernel (composite.hip) gives each 16-lane row one 4x4 block and reduces the 10 (mapping) / 7 (tracking)
// gradient terms of a (block, splat) step over the row with DPP butterflies: ~35 of its ~88 VALU instructions per step.  It is
// VALU-issue bound (profiles/r01_sq_counters.md), so the lever is instructions per (block, splat).  Here
//   * a wave covers a 16x8-pixel half tile = eight blocks; lane = (k, b, h): k = lane / 16 is the pixel ROW inside the block,
//     b = (lane % 16) / 2 the block, h = lane % 2 the left / right pixel pair of that row -- each lane owns 2 adjacent pixels, so
//     everything that depends on the splat and the row only (dy, conic x dy terms, the LDS reads of the record) is paid once per
//     2 pixels, and a wave step advances EIGHT block lists at once;
//   * the sum over the 16 pixels of a block = (in-lane sum over the 2 pixels) + (sum over the 4 rows k) + (sum over the 2 halves h).
//     The sum over k is exactly the K dimension of v_mfma_f32_16x16x4_f32 (A[i][k] from lane (i = lane % 16, k = lane / 16),
//     B[k][j] from lane (k, j = lane % 16)): with the per-lane value as B and a constant selector / weight matrix as A,
//         D[i][j] = sum_k A[i][k] * value(k, j)
//     lands, for column j = (b, h), the record floats 4g .. 4g+3 in lane (g, j) -- the lane that knows the record index of block
//     b's current entry -- so the store is one 16-byte write per lane.  The y-moments use A[i][k] = yt_k, yt_k^2 (yt = row offset
//     from the block centre): they come out about the BLOCK centre row and gather_records.h shifts them to the splat centre
//     (d0 = y_splat - y_centre: My = d0 M0 - My~, Mxy = d0 Mx - Mxy~, Myy = d0^2 M0 - 2 d0 My~ + Myy~); the x-moments use the true
//     dx of each pixel and need no shift.  One quad_perm add per result register joins the two halves.
//   7 (mapping) / 4 (tracking) MFMAs per wave step replace 4 x 35 DPP-butterfly instructions; the matrix pipe runs beside the VALU.
// The f32 MFMA is an exact k-ordered fmaf chain (bit-reproducible), so the backward pass stays deterministic.
//
// Everything else follows composite.hip: back-to-front traversal of the block lists built by sort_tile_body, T rebuilt by
// division, one running scalar for the colour behind, one plain store per (block, splat) record, zero records behind the
// deepest contributor.  Skip decisions use the same splat_power() as the forward pass.
#include <type_traits>
#include "composite_common.h"
#include "fused_api.h"
#include "loss_pixel.h"
#include "loss_tile.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define B2_QP_XOR1 0xB1   // quad_perm:[1,0,3,2]

template <int MODE>   // 1: mapping records [M0 Mx Mxx c0 | c1 c2 cz My~ | Mxy~ Myy~] (48-byte stride), 2: tracking [M0 Mx Mxx cz | My~ Mxy~ Myy~] (32-byte stride)
__global__ void __launch_bounds__(128)
composite_bwd2_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, const float* __restrict__ dL_dout,
                      float* __restrict__ dsub, int has_tl, TrackLoss tl, int dl_planes) {
  constexpr int C = 6;
  constexpr int RECF = MODE == 2 ? REC_TRACK_F : REC_MAP_F;    // packed records (composite_common.h)
  constexpr int NG = MODE == 2 ? 2 : 3;                 // float4 groups of a record that carry data (the last one partly)
  constexpr int ROW_CZ = MODE == 2 ? 3 : 6, ROW_MY = MODE == 2 ? 4 : 7, ROW_MXY = MODE == 2 ? 5 : 8, ROW_MYY = MODE == 2 ? 6 : 9;
  constexpr uint32_t CH = 16;                           // list entries staged per block and chunk
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T, cam.tilemap);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = lane >> 4, j = lane & 15, bq = j >> 1, h = j & 1;
  const int L = 8 * wv + bq;                            // block list: 4 * (8x8 sub-tile) + block inside it (sort_tile.h)
  const int sub = L >> 2, inb = L & 3;
  const int bx = (sub & 1) * 2 + (inb & 1), by = (sub >> 1) * 2 + (inb >> 1);
  const int px0 = (tile % cam.gx) * TILE + 4 * bx + 2 * h;
  const int py = (tile / cam.gx) * TILE + 4 * by + k;
  const bool in0 = px0 < cam.W && py < cam.H, in1 = px0 + 1 < cam.W && py < cam.H;
  const float pxf0 = (float)px0, pxf1 = (float)(px0 + 1), pyf = (float)py;
  uint32_t start, len_;
  tile_span(iv, tile, N_cap, start, len_);
  const uint32_t end = start + len_;
  const uint32_t len = end - start;
  const uint32_t count = len ? min(iv.subcount[NLIST * tile + L], len) : 0u;
  const uint2* __restrict__ list = b.sublist + (size_t)NLIST * start + (size_t)L * len;

  // [wave][buffer][field A|B|C][step * 8 + block]: lane-contiguous staging writes; a step's reads touch 8 consecutive 16-byte
  // slots (the 8 blocks), each broadcast to its 8 lanes -- conflict free
  __shared__ float4 stg[2][2][3][CH * 8];
  __shared__ uint32_t srec[2][2][CH * 8];
  __shared__ uint32_t s_todo[2][8];
  __shared__ unsigned long long s_list[2][8];           // element offset of each block's list inside b.sublist

  const size_t HW = (size_t)cam.H * cam.W;
  const size_t pix0 = (size_t)py * cam.W + px0;
  float Tf[2] = {in0 ? iv.final_T[pix0] : 0.f, in1 ? iv.final_T[pix0 + 1] : 0.f};
  uint32_t lastc[2] = {in0 ? iv.n_contrib[pix0] : 0u, in1 ? iv.n_contrib[pix0 + 1] : 0u};
  float dL[2][C];
  bool dl_done = false;
  if constexpr (MODE == 2) {
    if (has_tl) {
      // tracking loss folded in: dL/d(image) of these pixels from the finished sums (what loss_grad_kernel would have written)
      const float l1s = tl.defer_scale ? tl.cfg.w_l1 / 3.f : loss_l1_scale(tl.cfg, tl.sums);   // deferred: 1/n applied to the pose gradient
#pragma unroll
      for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int ch = 0; ch < C; ch++) dL[p][ch] = 0.f;
        if (p == 0 ? in0 : in1) {
          const size_t pix = pix0 + p;
          const float sil = tl.out[4 * HW + pix];
          const bool smask = sil > tl.cfg.sil_thr;
#pragma unroll
          for (int ch = 0; ch < 3; ch++) dL[p][ch] = loss_px_l1_grad(tl.cfg, tl.out[ch * HW + pix], tl.gt[ch * HW + pix], smask, l1s);
          if (tl.cfg.w_pearson != 0.f) dL[p][3] = loss_px_pearson_grad(tl.cfg, sil, tl.out[3 * HW + pix], tl.ref[pix], tl.sums);
        }
      }
      if (!tl.defer_scale && tile == 0 && tid == 0 && tl.loss4) loss_scalars(tl.cfg, tl.sums, HW, tl.loss4);
      dl_done = true;
    }
  }
  float Tf_bg[2] = {0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 2; p++) {
    float bg_dot = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
      if (!dl_done) dL[p][ch] = ((p == 0 ? in0 : in1) && ch < dl_planes) ? dL_dout[ch * HW + pix0 + p] : 0.f;
      if (ch < 3) bg_dot += cam.bg[ch] * dL[p][ch];
    }
    Tf_bg[p] = Tf[p] * bg_dot;
  }
  float Tr[2] = {Tf[0], Tf[1]};
  float behind[2] = {0.f, 0.f};      // (colour accumulated behind the current list position) . dL, per pixel

  // nothing behind the deepest contributor of any pixel of the block matters: todo = max over the block's 16 pixels
  // (2 in this lane, the other half one lane over, the other rows 16 / 32 lanes over)
  uint32_t todo = max(lastc[0], lastc[1]);
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 1, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 16, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 32, 64));
  todo = min(todo, count);
  uint32_t maxtodo = todo;
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 2, 64));
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 4, 64));
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 8, 64));
  maxtodo = __builtin_amdgcn_readfirstlane(maxtodo);
  if (k == 0 && h == 0) {
    s_todo[wv][bq] = todo;
    s_list[wv][bq] = (unsigned long long)NLIST * start + (unsigned long long)L * len;
  }
  // entries behind `todo` receive no gradient: their records are zero (8 lanes per block)
  {
    const uint32_t q8 = (uint32_t)(k * 2 + h);
    for (uint32_t e = todo + q8; e < count; e += 8) {
      zero_record<RECF>(dsub + (size_t)list[e].y * RECF);
    }
  }
  __syncthreads();                   // s_todo / s_list visible (both waves take the same path up to here)
  if (maxtodo == 0) return;          // wave-uniform

  // selector / weight matrices of the MFMAs: lane (i = lane % 16, kk = lane / 16) supplies A[i][kk]
  const float yt = (float)k - 1.5f;  // this lane group's row offset from the block centre
  const int ai = lane & 15;
  const float aU0 = ai == 0 ? 1.f : (ai == ROW_MY ? yt : (ai == ROW_MYY ? yt * yt : 0.f));
  const float aU1 = ai == 1 ? 1.f : (ai == ROW_MXY ? yt : 0.f);
  const float aU2 = ai == 2 ? 1.f : 0.f;
  const float aCZ = ai == ROW_CZ ? 1.f : 0.f;
  const float aP0 = ai == 3 ? 1.f : 0.f, aP1 = ai == 4 ? 1.f : 0.f, aP2 = ai == 5 ? 1.f : 0.f;   // mapping only

  // staging: this lane loads the entries of slots `lane` and `lane + 64` of a chunk: slot = step * 8 + block
  const int sb0 = lane & 7, ss0 = lane >> 3, ss1 = ss0 + 8;     // both slots belong to block sb0, steps ss0 and ss0 + 8
  const uint32_t stodo = s_todo[wv][sb0];
  const uint2* __restrict__ slist = b.sublist + s_list[wv][sb0];
  auto entry_at = [&](uint32_t pos_from_back) -> uint2 {           // traversal position -> list entry (back to front)
    return pos_from_back < stodo ? slist[stodo - 1u - pos_from_back] : make_uint2(0u, 0u);
  };
  {
    const uint2 e0 = entry_at((uint32_t)ss0), e1 = entry_at((uint32_t)ss1);
    const SplatRec r0 = load_rec<C>(g.splat, e0.x, (uint32_t)ss0 < stodo), r1 = load_rec<C>(g.splat, e1.x, (uint32_t)ss1 < stodo);
    stg[wv][0][0][lane] = r0.A; stg[wv][0][1][lane] = r0.B; stg[wv][0][2][lane] = r0.C; srec[wv][0][lane] = e0.y;
    stg[wv][0][0][lane + 64] = r1.A; stg[wv][0][1][lane + 64] = r1.B; stg[wv][0][2][lane + 64] = r1.C; srec[wv][0][lane + 64] = e1.y;
  }
  uint2 en0 = entry_at(CH + (uint32_t)ss0), en1 = entry_at(CH + (uint32_t)ss1);
  int cur = 0;

  // The SLAM losses leave the silhouette and depth^2 channels without gradient (dL[4] = dL[5] = 0): a wave that sees only
  // zeros there runs a loop instance with those terms removed (exact: they would multiply by zero).
  const bool z45_wave = __ballot(dL[0][4] != 0.f || dL[0][5] != 0.f || dL[1][4] != 0.f || dL[1][5] != 0.f) == 0ull;
  auto run_chunks = [&](auto z45_tag) {
    constexpr bool Z45 = decltype(z45_tag)::value;
    for (uint32_t base = 0; base < maxtodo; base += CH, cur ^= 1) {
      // gathers for the following chunks are issued before this one is touched; they land while it is processed
      const SplatRec rn0 = load_rec<C>(g.splat, en0.x, base + CH + (uint32_t)ss0 < stodo);
      const SplatRec rn1 = load_rec<C>(g.splat, en1.x, base + CH + (uint32_t)ss1 < stodo);
      const uint2 enn0 = entry_at(base + 2 * CH + (uint32_t)ss0), enn1 = entry_at(base + 2 * CH + (uint32_t)ss1);
      const float4 (*wS)[CH * 8] = stg[wv][cur];
      const uint32_t* wR = srec[wv][cur];
      __builtin_amdgcn_wave_barrier();
      const int cnt = __builtin_amdgcn_readfirstlane((int)min(CH, maxtodo - base));
      auto step = [&](const float4& A, const float4& B, const float4& Cc, const uint32_t rec, const int s) {
        const bool blk_on = base + (uint32_t)s < todo;                 // this block still has an entry at this step
        const uint32_t pos = todo - 1u - (base + (uint32_t)s);          // 0-based index in the block's list (garbage when !blk_on)
        const float dy = A.y - pyf;
        float u_[2], udx_[2], udxx_[2], w_[2];
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, cz = 0.f;
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const float dx = A.x - (p == 0 ? pxf0 : pxf1);
          const float power = splat_power(dx, dy, A.z, A.w, B.x);
          const float G = __expf(power);
          const float alpha = fminf(0.99f, B.y * G);
          const bool valid = blk_on && (pos < lastc[p]) && !(power > 0.f) && !(alpha < ALPHA_MIN);
          const float a_eff = valid ? alpha : 0.f;
          const float G_eff = valid ? G : 0.f;
          const float r = __builtin_amdgcn_rcpf(1.f - a_eff);
          Tr[p] *= r;                                                   // transmittance in front of this splat
          const float w = a_eff * Tr[p];
          // dL/dalpha needs sum_ch (c_ch - behind_ch) dL_ch: the dL-weighted colour behind is ONE running scalar
          float qd = B.z * dL[p][0];
          qd = fmaf(B.w, dL[p][1], qd); qd = fmaf(Cc.x, dL[p][2], qd); qd = fmaf(Cc.y, dL[p][3], qd);
          if (!Z45) { qd = fmaf(Cc.z, dL[p][4], qd); qd = fmaf(Cc.w, dL[p][5], qd); }
          const float diff = qd - behind[p];
          behind[p] = fmaf(a_eff, diff, behind[p]);
          const float dLa = diff * Tr[p] - Tf_bg[p] * r;
          const float u = B.y * dLa * G_eff;                           // dL/dG * G: its moments give d/dxy and d/dconic
          u_[p] = u; udx_[p] = u * dx; udxx_[p] = udx_[p] * dx; w_[p] = w;
          if (MODE == 1) { p0 = fmaf(w, dL[p][0], p0); p1 = fmaf(w, dL[p][1], p1); p2 = fmaf(w, dL[p][2], p2); }
          cz = fmaf(w, Z45 ? dL[p][3] : fmaf(2.f * Cc.y, dL[p][5], dL[p][3]), cz);   // d/dz of the [z, 1, z^2] bundle, chained here
        }
        const float U0 = u_[0] + u_[1], U1 = udx_[0] + udx_[1], U2 = udxx_[0] + udxx_[1];
        f32x4 D = {0.f, 0.f, 0.f, 0.f};
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aU0, U0, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aU1, U1, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aU2, U2, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aCZ, cz, D, 0, 0, 0);
        if (MODE == 1) {
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(aP0, p0, D, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(aP1, p1, D, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(aP2, p2, D, 0, 0, 0);
        }
        // lane (gq = lane / 16, column (block, half)) holds record floats 4 gq .. 4 gq + 3 of its half: add the other half
        float4 R;
        R.x = D[0] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D[0]), B2_QP_XOR1, 0xf, 0xf, true));
        R.y = D[1] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D[1]), B2_QP_XOR1, 0xf, 0xf, true));
        R.z = D[2] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D[2]), B2_QP_XOR1, 0xf, 0xf, true));
        R.w = D[3] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(D[3]), B2_QP_XOR1, 0xf, 0xf, true));
        // the block's lanes are the only writers of the (block, splat) record: NG lanes store 16 bytes each
        if (blk_on && h == 0 && k < NG) st_part(dsub + (size_t)rec * RECF + 4 * k, R, min(4, RECF - 4 * k));
      };
      // two register sets used alternately: the next step's LDS reads are in flight while the current one is evaluated
      float4 A0 = wS[0][bq], B0 = wS[1][bq], C0 = wS[2][bq];
      uint32_t t0 = wR[bq];
      for (int s = 0; s < cnt; s += 2) {
        const int s1 = s + 1 < cnt ? s + 1 : s;
        const float4 A1 = wS[0][s1 * 8 + bq], B1 = wS[1][s1 * 8 + bq], C1 = wS[2][s1 * 8 + bq];
        const uint32_t t1 = wR[s1 * 8 + bq];
        step(A0, B0, C0, t0, s);
        if (s + 1 < cnt) {
          const int s2 = s + 2 < cnt ? s + 2 : s1;
          A0 = wS[0][s2 * 8 + bq]; B0 = wS[1][s2 * 8 + bq]; C0 = wS[2][s2 * 8 + bq];
          t0 = wR[s2 * 8 + bq];
          step(A1, B1, C1, t1, s1);
        }
      }
      stg[wv][cur ^ 1][0][lane] = rn0.A; stg[wv][cur ^ 1][1][lane] = rn0.B; stg[wv][cur ^ 1][2][lane] = rn0.C; srec[wv][cur ^ 1][lane] = en0.y;
      stg[wv][cur ^ 1][0][lane + 64] = rn1.A; stg[wv][cur ^ 1][1][lane + 64] = rn1.B; stg[wv][cur ^ 1][2][lane + 64] = rn1.C;
      srec[wv][cur ^ 1][lane + 64] = en1.y;
      en0 = enn0; en1 = enn1;
    }
  };
  if (z45_wave) run_chunks(std::true_type{});
  else run_chunks(std::false_type{});
}

// ---- fourth generation: two pixels per lane like the second, but the block's eight lanes are CONTIGUOUS (lane = 8 * block + 2 * row +
// half), so the sum over a block is three DPP steps (quad_perm xor 1, xor 2, row_half_mirror) per value instead of a matrix-core
// pass: 30 (mapping) / 21 (tracking) v_add_f32_dpp per wave step for EIGHT records, against ~34 per four records in the first
// generation, and the y-moments are taken about the splat centre directly (dy is a per-lane constant of the step): the records
// are the first generation's, no shift in the gather.
template <int MODE>   // 1: mapping records [M0 Mx Mxx c0 | c1 c2 cz My | Mxy Myy], 2: tracking [M0 Mx Mxx cz | My Mxy Myy]
__global__ void __launch_bounds__(128)
composite_bwd4_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, const float* __restrict__ dL_dout,
                      float* __restrict__ dsub, int has_tl, TrackLoss tl, int dl_planes) {
  constexpr int C = 6;
  constexpr int RECF = MODE == 2 ? REC_TRACK_F : REC_MAP_F;    // packed records (composite_common.h)
  constexpr uint32_t CH = 16;                           // list entries staged per block and chunk
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T, cam.tilemap);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int bq = lane >> 3, k = (lane >> 1) & 3, h = lane & 1;
  const int L = 8 * wv + bq;                            // block list: 4 * (8x8 sub-tile) + block inside it (sort_tile.h)
  const int sub = L >> 2, inb = L & 3;
  const int bx = (sub & 1) * 2 + (inb & 1), by = (sub >> 1) * 2 + (inb >> 1);
  const int px0 = (tile % cam.gx) * TILE + 4 * bx + 2 * h;
  const int py = (tile / cam.gx) * TILE + 4 * by + k;
  const bool in0 = px0 < cam.W && py < cam.H, in1 = px0 + 1 < cam.W && py < cam.H;
  const float pxf0 = (float)px0, pxf1 = (float)(px0 + 1), pyf = (float)py;
  uint32_t start, len_;
  tile_span(iv, tile, N_cap, start, len_);
  const uint32_t end = start + len_;
  const uint32_t len = end - start;
  const uint32_t count = len ? min(iv.subcount[NLIST * tile + L], len) : 0u;
  const uint2* __restrict__ list = b.sublist + (size_t)NLIST * start + (size_t)L * len;

  // [wave][buffer][field A|B|C][step * 8 + block]: lane-contiguous staging writes; a step's reads touch 8 consecutive 16-byte
  // slots (the 8 blocks), each broadcast to its 8 lanes -- conflict free
  __shared__ float4 stg[2][2][3][CH * 8];
  __shared__ uint32_t srec[2][2][CH * 8];
  __shared__ uint32_t s_todo[2][8];
  __shared__ unsigned long long s_list[2][8];           // element offset of each block's list inside b.sublist

  const size_t HW = (size_t)cam.H * cam.W;
  const size_t pix0 = (size_t)py * cam.W + px0;
  float Tf[2] = {in0 ? iv.final_T[pix0] : 0.f, in1 ? iv.final_T[pix0 + 1] : 0.f};
  uint32_t lastc[2] = {in0 ? iv.n_contrib[pix0] : 0u, in1 ? iv.n_contrib[pix0 + 1] : 0u};
  float dL[2][C];
  bool dl_done = false;
  if constexpr (MODE == 2) {
    if (has_tl) {
      // tracking loss folded in: dL/d(image) of these pixels from the finished sums (what loss_grad_kernel would have written)
      const float l1s = tl.defer_scale ? tl.cfg.w_l1 / 3.f : loss_l1_scale(tl.cfg, tl.sums);   // deferred: 1/n applied to the pose gradient
#pragma unroll
      for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int ch = 0; ch < C; ch++) dL[p][ch] = 0.f;
        if (p == 0 ? in0 : in1) {
          const size_t pix = pix0 + p;
          const float sil = tl.out[4 * HW + pix];
          const bool smask = sil > tl.cfg.sil_thr;
#pragma unroll
          for (int ch = 0; ch < 3; ch++) dL[p][ch] = loss_px_l1_grad(tl.cfg, tl.out[ch * HW + pix], tl.gt[ch * HW + pix], smask, l1s);
          if (tl.cfg.w_pearson != 0.f) dL[p][3] = loss_px_pearson_grad(tl.cfg, sil, tl.out[3 * HW + pix], tl.ref[pix], tl.sums);
        }
      }
      if (!tl.defer_scale && tile == 0 && tid == 0 && tl.loss4) loss_scalars(tl.cfg, tl.sums, HW, tl.loss4);
      dl_done = true;
    }
  }
  float Tf_bg[2] = {0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 2; p++) {
    float bg_dot = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
      if (!dl_done) dL[p][ch] = ((p == 0 ? in0 : in1) && ch < dl_planes) ? dL_dout[ch * HW + pix0 + p] : 0.f;
      if (ch < 3) bg_dot += cam.bg[ch] * dL[p][ch];
    }
    Tf_bg[p] = Tf[p] * bg_dot;
  }
  float Tr[2] = {Tf[0], Tf[1]};
  float behind[2] = {0.f, 0.f};      // (colour accumulated behind the current list position) . dL, per pixel

  // nothing behind the deepest contributor of any pixel of the block matters: todo = max over the block's 16 pixels
  // (2 in this lane, the other half one lane over, the other rows 16 / 32 lanes over)
  uint32_t todo = max(lastc[0], lastc[1]);
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 1, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 2, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 4, 64));
  todo = min(todo, count);
  uint32_t maxtodo = todo;
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 8, 64));
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 16, 64));
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 32, 64));
  maxtodo = __builtin_amdgcn_readfirstlane(maxtodo);
  if ((lane & 7) == 0) {
    s_todo[wv][bq] = todo;
    s_list[wv][bq] = (unsigned long long)NLIST * start + (unsigned long long)L * len;
  }
  // entries behind `todo` receive no gradient: their records are zero (8 lanes per block)
  {
    const uint32_t q8 = (uint32_t)(lane & 7);
    for (uint32_t e = todo + q8; e < count; e += 8) {
      zero_record<RECF>(dsub + (size_t)list[e].y * RECF);
    }
  }
  __syncthreads();                   // s_todo / s_list visible (both waves take the same path up to here)
  if (maxtodo == 0) return;          // wave-uniform

  // staging: this lane loads the entries of slots `lane` and `lane + 64` of a chunk: slot = step * 8 + block
  const int sb0 = lane & 7, ss0 = lane >> 3, ss1 = ss0 + 8;     // both slots belong to block sb0, steps ss0 and ss0 + 8
  const uint32_t stodo = s_todo[wv][sb0];
  const uint2* __restrict__ slist = b.sublist + s_list[wv][sb0];
  auto entry_at = [&](uint32_t pos_from_back) -> uint2 {           // traversal position -> list entry (back to front)
    return pos_from_back < stodo ? slist[stodo - 1u - pos_from_back] : make_uint2(0u, 0u);
  };
  {
    const uint2 e0 = entry_at((uint32_t)ss0), e1 = entry_at((uint32_t)ss1);
    const SplatRec r0 = load_rec<C>(g.splat, e0.x, (uint32_t)ss0 < stodo), r1 = load_rec<C>(g.splat, e1.x, (uint32_t)ss1 < stodo);
    stg[wv][0][0][lane] = r0.A; stg[wv][0][1][lane] = r0.B; stg[wv][0][2][lane] = r0.C; srec[wv][0][lane] = e0.y;
    stg[wv][0][0][lane + 64] = r1.A; stg[wv][0][1][lane + 64] = r1.B; stg[wv][0][2][lane + 64] = r1.C; srec[wv][0][lane + 64] = e1.y;
  }
  uint2 en0 = entry_at(CH + (uint32_t)ss0), en1 = entry_at(CH + (uint32_t)ss1);
  int cur = 0;

  // The SLAM losses leave the silhouette and depth^2 channels without gradient (dL[4] = dL[5] = 0): a wave that sees only
  // zeros there runs a loop instance with those terms removed (exact: they would multiply by zero).
  const bool z45_wave = __ballot(dL[0][4] != 0.f || dL[0][5] != 0.f || dL[1][4] != 0.f || dL[1][5] != 0.f) == 0ull;
  auto run_chunks = [&](auto z45_tag) {
    constexpr bool Z45 = decltype(z45_tag)::value;
    for (uint32_t base = 0; base < maxtodo; base += CH, cur ^= 1) {
      // gathers for the following chunks are issued before this one is touched; they land while it is processed
      const SplatRec rn0 = load_rec<C>(g.splat, en0.x, base + CH + (uint32_t)ss0 < stodo);
      const SplatRec rn1 = load_rec<C>(g.splat, en1.x, base + CH + (uint32_t)ss1 < stodo);
      const uint2 enn0 = entry_at(base + 2 * CH + (uint32_t)ss0), enn1 = entry_at(base + 2 * CH + (uint32_t)ss1);
      const float4 (*wS)[CH * 8] = stg[wv][cur];
      const uint32_t* wR = srec[wv][cur];
      __builtin_amdgcn_wave_barrier();
      const int cnt = __builtin_amdgcn_readfirstlane((int)min(CH, maxtodo - base));
      auto step = [&](const float4& A, const float4& B, const float4& Cc, const uint32_t rec, const int s) {
        const bool blk_on = base + (uint32_t)s < todo;                 // this block still has an entry at this step
        const uint32_t pos = todo - 1u - (base + (uint32_t)s);          // 0-based index in the block's list (garbage when !blk_on)
        const float dy = A.y - pyf;
        float u_[2], udx_[2], udxx_[2], w_[2];
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, cz = 0.f;
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const float dx = A.x - (p == 0 ? pxf0 : pxf1);
          const float power = splat_power(dx, dy, A.z, A.w, B.x);
          const float G = __expf(power);
          const float alpha = fminf(0.99f, B.y * G);
          const bool valid = blk_on && (pos < lastc[p]) && !(power > 0.f) && !(alpha < ALPHA_MIN);
          const float a_eff = valid ? alpha : 0.f;
          const float G_eff = valid ? G : 0.f;
          const float r = __builtin_amdgcn_rcpf(1.f - a_eff);
          Tr[p] *= r;                                                   // transmittance in front of this splat
          const float w = a_eff * Tr[p];
          // dL/dalpha needs sum_ch (c_ch - behind_ch) dL_ch: the dL-weighted colour behind is ONE running scalar
          float qd = B.z * dL[p][0];
          qd = fmaf(B.w, dL[p][1], qd); qd = fmaf(Cc.x, dL[p][2], qd); qd = fmaf(Cc.y, dL[p][3], qd);
          if (!Z45) { qd = fmaf(Cc.z, dL[p][4], qd); qd = fmaf(Cc.w, dL[p][5], qd); }
          const float diff = qd - behind[p];
          behind[p] = fmaf(a_eff, diff, behind[p]);
          const float dLa = diff * Tr[p] - Tf_bg[p] * r;
          const float u = B.y * dLa * G_eff;                           // dL/dG * G: its moments give d/dxy and d/dconic
          u_[p] = u; udx_[p] = u * dx; udxx_[p] = udx_[p] * dx; w_[p] = w;
          if (MODE == 1) { p0 = fmaf(w, dL[p][0], p0); p1 = fmaf(w, dL[p][1], p1); p2 = fmaf(w, dL[p][2], p2); }
          cz = fmaf(w, Z45 ? dL[p][3] : fmaf(2.f * Cc.y, dL[p][5], dL[p][3]), cz);   // d/dz of the [z, 1, z^2] bundle, chained here
        }
        const float U0 = u_[0] + u_[1], U1 = udx_[0] + udx_[1], U2 = udxx_[0] + udxx_[1];
        // y-moments about the splat centre: dy is this lane's constant of the step
        float vals[MODE == 1 ? 10 : 7];
        if constexpr (MODE == 1) {
          vals[0] = U0; vals[1] = U1; vals[2] = U2; vals[3] = p0; vals[4] = p1; vals[5] = p2; vals[6] = cz;
          vals[7] = U0 * dy; vals[8] = U1 * dy; vals[9] = vals[7] * dy;
        } else {
          vals[0] = U0; vals[1] = U1; vals[2] = U2; vals[3] = cz; vals[4] = U0 * dy; vals[5] = U1 * dy; vals[6] = vals[4] * dy;
        }
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int q = 0; q < (MODE == 1 ? 10 : 7); q++) {
            float v = vals[q];
            v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
            v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
            v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
            vals[q] = v;
          }
        }
        // every lane of the block holds the record: its first lane stores it (the only writer of the (block, splat) record)
        if (blk_on && (lane & 7) == 0) {
          float* r = dsub + (size_t)rec * RECF;
          const float4 R0 = make_float4(vals[0], vals[1], vals[2], vals[3]);
          st_part(r, R0, 4);
          if constexpr (MODE == 1) {
            st_part(r + 4, make_float4(vals[4], vals[5], vals[6], vals[7]), 4);
            st_part(r + 8, make_float4(vals[8], vals[9], 0.f, 0.f), 2);
          } else {
            st_part(r + 4, make_float4(vals[4], vals[5], vals[6], 0.f), 3);
          }
        }
      };
      // two register sets used alternately: the next step's LDS reads are in flight while the current one is evaluated
      float4 A0 = wS[0][bq], B0 = wS[1][bq], C0 = wS[2][bq];
      uint32_t t0 = wR[bq];
      for (int s = 0; s < cnt; s += 2) {
        const int s1 = s + 1 < cnt ? s + 1 : s;
        const float4 A1 = wS[0][s1 * 8 + bq], B1 = wS[1][s1 * 8 + bq], C1 = wS[2][s1 * 8 + bq];
        const uint32_t t1 = wR[s1 * 8 + bq];
        step(A0, B0, C0, t0, s);
        if (s + 1 < cnt) {
          const int s2 = s + 2 < cnt ? s + 2 : s1;
          A0 = wS[0][s2 * 8 + bq]; B0 = wS[1][s2 * 8 + bq]; C0 = wS[2][s2 * 8 + bq];
          t0 = wR[s2 * 8 + bq];
          step(A1, B1, C1, t1, s1);
        }
      }
      stg[wv][cur ^ 1][0][lane] = rn0.A; stg[wv][cur ^ 1][1][lane] = rn0.B; stg[wv][cur ^ 1][2][lane] = rn0.C; srec[wv][cur ^ 1][lane] = en0.y;
      stg[wv][cur ^ 1][0][lane + 64] = rn1.A; stg[wv][cur ^ 1][1][lane + 64] = rn1.B; stg[wv][cur ^ 1][2][lane + 64] = rn1.C;
      srec[wv][cur ^ 1][lane + 64] = en1.y;
      en0 = enn0; en1 = enn1;
    }
  };
  if (z45_wave) run_chunks(std::true_type{});
  else run_chunks(std::false_type{});
}


// ---- third generation: ONE pixel per lane (the first generation's 4800 waves at 640x480: 4.7 per SIMD cover the in-order
// stalls that 2.3 could not), the matrix-core block reduction of the second.  A wave = an 8x8 sub-tile = four blocks; lane =
// (k, b, x): k = lane / 16 the pixel row inside the block, b = (lane % 16) / 4 the block, x = lane % 4 the pixel column.  The MFMA
// sums over k; the four columns x of a block sit in one quad and are joined with two quad_perm adds per result register.
template <int MODE>
__global__ void __launch_bounds__(256, 5)   // 5 waves per SIMD: the 4800 waves of a 640x480 view resident at once
composite_bwd3_kernel(CamDev cam, GeomView g, ImageView iv, BinView b, uint32_t N_cap, const float* __restrict__ dL_dout,
                      float* __restrict__ dsub, int has_tl, TrackLoss tl, int dl_planes) {
  constexpr int C = 6;
  constexpr int RECF = MODE == 2 ? REC_TRACK_F : REC_MAP_F;    // packed records (composite_common.h)
  constexpr int NG = MODE == 2 ? 2 : 3;                 // float4 groups of a record that carry data (the last one partly)
  constexpr int ROW_CZ = MODE == 2 ? 3 : 6, ROW_MY = MODE == 2 ? 4 : 7, ROW_MXY = MODE == 2 ? 5 : 8, ROW_MYY = MODE == 2 ? 6 : 9;
  constexpr uint32_t CH = 16;                           // list entries staged per block and chunk
  const int T = cam.gx * cam.gy;
  const int tile = xcd_tile(blockIdx.x, T, cam.tilemap);
  if (tile >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k = lane >> 4, j = lane & 15, bq = j >> 2, x = j & 3;
  const int L = 4 * wv + bq;                            // block list: 4 * (8x8 sub-tile) + block inside it (sort_tile.h)
  const int px = (tile % cam.gx) * TILE + (wv & 1) * 8 + (bq & 1) * 4 + x;
  const int py = (tile / cam.gx) * TILE + (wv >> 1) * 8 + (bq >> 1) * 4 + k;
  const bool inside = px < cam.W && py < cam.H;
  const float pxf = (float)px, pyf = (float)py;
  uint32_t start, len_;
  tile_span(iv, tile, N_cap, start, len_);
  const uint32_t end = start + len_;
  const uint32_t len = end - start;
  const uint32_t count = len ? min(iv.subcount[NLIST * tile + L], len) : 0u;
  const uint2* __restrict__ list = b.sublist + (size_t)NLIST * start + (size_t)L * len;

  // [wave][buffer][field A|B|C][block * 16 + entry] (+ the record index): lane-contiguous staging writes, block-uniform reads.
  // The folded mapping-loss gradient pass (loss_tile.h) uses the same memory first.
  __shared__ __align__(16) unsigned char smem_raw[sizeof(float4) * 4 * 2 * 3 * 64 + sizeof(uint32_t) * 4 * 2 * 64];
  static_assert(sizeof(smem_raw) >= sizeof(LossGradSmem) + 4 * 256 * sizeof(float), "LDS union too small for the loss pass");
  float4 (*stg)[2][3][64] = (float4 (*)[2][3][64])smem_raw;
  uint32_t (*srec)[2][64] = (uint32_t (*)[2][64])(smem_raw + sizeof(float4) * 4 * 2 * 3 * 64);
  __shared__ uint32_t s_todo[4][4];
  __shared__ unsigned long long s_list[4][4];

  const size_t HW = (size_t)cam.H * cam.W;
  const size_t pix = (size_t)py * cam.W + px;
  const float Tf = inside ? iv.final_T[pix] : 0.f;
  const uint32_t lastc = inside ? iv.n_contrib[pix] : 0u;
  float dL[C];
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
      if (!tl.defer_scale && tile == 0 && tid == 0 && tl.loss4) loss_scalars(tl.cfg, tl.sums, HW, tl.loss4);
      dl_done = true;
    }
  }
  if constexpr (MODE == 1) {
    if (has_tl) {
      // mapping loss folded in (see composite_bwd_kernel): the gradient-image pass of this tile, in raster order, then each
      // lane picks its own pixel up through LDS
      LossGradSmem& lsm = *(LossGradSmem*)smem_raw;
      float* exch = (float*)(smem_raw + sizeof(LossGradSmem));
      float g4[4];
      bool in_raster;
      loss_grad_tile(tl.cfg, tl.out, tl.gt, tl.ref, tl.dmaps, tl.sums, tile, cam.gx, lsm, g4, in_raster);
#pragma unroll
      for (int ch = 0; ch < 4; ch++) exch[ch * 256 + tid] = g4[ch];
      __syncthreads();
      const int lx = (wv & 1) * 8 + (bq & 1) * 4 + x, ly = (wv >> 1) * 8 + (bq >> 1) * 4 + k;
#pragma unroll
      for (int ch = 0; ch < C; ch++) dL[ch] = ch < 4 ? exch[ch * 256 + ly * 16 + lx] : 0.f;
      __syncthreads();      // the staging buffers reuse this memory
      dl_done = true;
    }
  }
  float bg_dot = 0.f;
#pragma unroll
  for (int ch = 0; ch < C; ch++) {
    if (!dl_done) dL[ch] = (inside && ch < dl_planes) ? dL_dout[ch * HW + pix] : 0.f;
    if (ch < 3) bg_dot += cam.bg[ch] * dL[ch];
  }
  const float Tf_bg = Tf * bg_dot;
  float Tr = Tf;
  float behind = 0.f;                // (colour accumulated behind the current list position) . dL

  // nothing behind the deepest contributor of any pixel of the block matters: todo = max over the block's 16 pixels
  // (the quad's 4 columns, then the 4 rows 16 / 32 lanes over)
  uint32_t todo = lastc;
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 1, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 2, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 16, 64));
  todo = max(todo, (uint32_t)__shfl_xor((int)todo, 32, 64));
  todo = min(todo, count);
  uint32_t maxtodo = todo;
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 4, 64));
  maxtodo = max(maxtodo, (uint32_t)__shfl_xor((int)maxtodo, 8, 64));
  maxtodo = __builtin_amdgcn_readfirstlane(maxtodo);
  if (k == 0 && x == 0) {
    s_todo[wv][bq] = todo;
    s_list[wv][bq] = (unsigned long long)NLIST * start + (unsigned long long)L * len;
  }
  // entries behind `todo` receive no gradient: their records are zero (16 lanes per block)
  {
    const uint32_t q16 = (uint32_t)(k * 4 + x);
    for (uint32_t e = todo + q16; e < count; e += 16) {
      zero_record<RECF>(dsub + (size_t)list[e].y * RECF);
    }
  }
  __syncthreads();                   // s_todo / s_list visible (every wave takes the same path up to here)
  if (maxtodo == 0) return;          // wave-uniform

  // selector / weight matrices of the MFMAs: lane (i = lane % 16, kk = lane / 16) supplies A[i][kk]
  const float yt = (float)k - 1.5f;  // this lane group's row offset from the block centre
  const int ai = lane & 15;
  const float aU0 = ai == 0 ? 1.f : (ai == ROW_MY ? yt : (ai == ROW_MYY ? yt * yt : 0.f));
  const float aU1 = ai == 1 ? 1.f : (ai == ROW_MXY ? yt : 0.f);
  const float aU2 = ai == 2 ? 1.f : 0.f;
  const float aCZ = ai == ROW_CZ ? 1.f : 0.f;
  const float aP0 = ai == 3 ? 1.f : 0.f, aP1 = ai == 4 ? 1.f : 0.f, aP2 = ai == 5 ? 1.f : 0.f;   // mapping only

  // staging: lane (sb = lane / 16, sq = lane % 16) loads entry sq of block sb's chunk
  const int sb = lane >> 4, sq = lane & 15;
  const uint32_t stodo = s_todo[wv][sb];
  const uint2* __restrict__ slist = b.sublist + s_list[wv][sb];
  auto entry_at = [&](uint32_t pos_from_back) -> uint2 {           // traversal position -> list entry (back to front)
    return pos_from_back < stodo ? slist[stodo - 1u - pos_from_back] : make_uint2(0u, 0u);
  };
  {
    const uint2 e0 = entry_at((uint32_t)sq);
    const SplatRec r0 = load_rec<C>(g.splat, e0.x, (uint32_t)sq < stodo);
    stg[wv][0][0][lane] = r0.A; stg[wv][0][1][lane] = r0.B; stg[wv][0][2][lane] = r0.C; srec[wv][0][lane] = e0.y;
  }
  uint2 en0 = entry_at(CH + (uint32_t)sq);
  int cur = 0;

  const bool z45_wave = __ballot(dL[4] != 0.f || dL[5] != 0.f) == 0ull;
  auto run_chunks = [&](auto z45_tag) {
    constexpr bool Z45 = decltype(z45_tag)::value;
    for (uint32_t base = 0; base < maxtodo; base += CH, cur ^= 1) {
      const SplatRec rn0 = load_rec<C>(g.splat, en0.x, base + CH + (uint32_t)sq < stodo);
      const uint2 enn0 = entry_at(base + 2 * CH + (uint32_t)sq);
      const float4 (*wS)[64] = stg[wv][cur];
      const uint32_t* wR = srec[wv][cur];
      const int r16 = bq * 16;
      __builtin_amdgcn_wave_barrier();
      const int cnt = __builtin_amdgcn_readfirstlane((int)min(CH, maxtodo - base));
      auto step = [&](const float4& A, const float4& B, const float4& Cc, const uint32_t rec, const int s) {
        const bool blk_on = base + (uint32_t)s < todo;                 // this block still has an entry at this step
        const uint32_t pos = todo - 1u - (base + (uint32_t)s);          // 0-based index in the block's list (garbage when !blk_on)
        const float dx = A.x - pxf, dy = A.y - pyf;
        const float power = splat_power(dx, dy, A.z, A.w, B.x);
        const float G = __expf(power);
        const float alpha = fminf(0.99f, B.y * G);
        const bool valid = blk_on && (pos < lastc) && !(power > 0.f) && !(alpha < ALPHA_MIN);
        const float a_eff = valid ? alpha : 0.f;
        const float G_eff = valid ? G : 0.f;
        const float r = __builtin_amdgcn_rcpf(1.f - a_eff);
        Tr *= r;                                                        // transmittance in front of this splat
        const float w = a_eff * Tr;
        float qd = B.z * dL[0];
        qd = fmaf(B.w, dL[1], qd); qd = fmaf(Cc.x, dL[2], qd); qd = fmaf(Cc.y, dL[3], qd);
        if (!Z45) { qd = fmaf(Cc.z, dL[4], qd); qd = fmaf(Cc.w, dL[5], qd); }
        const float diff = qd - behind;
        behind = fmaf(a_eff, diff, behind);
        const float dLa = diff * Tr - Tf_bg * r;
        const float u = B.y * dLa * G_eff;                             // dL/dG * G: its moments give d/dxy and d/dconic
        const float udx = u * dx;
        f32x4 D = {0.f, 0.f, 0.f, 0.f};
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aU0, u, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aU1, udx, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aU2, udx * dx, D, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(aCZ, w * (Z45 ? dL[3] : fmaf(2.f * Cc.y, dL[5], dL[3])), D, 0, 0, 0);
        if (MODE == 1) {
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(aP0, w * dL[0], D, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(aP1, w * dL[1], D, 0, 0, 0);
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(aP2, w * dL[2], D, 0, 0, 0);
        }
        // lane (gq = lane / 16, column (block, x)) holds record floats 4 gq .. 4 gq + 3 of its pixel column: join the quad
        float4 R;
        float* Rp = (float*)&R;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float v = D[c];
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
          v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm:[2,3,0,1]
          Rp[c] = v;
        }
        // the block's lanes are the only writers of the (block, splat) record: NG lanes store 16 bytes each
        if (blk_on && x == 0 && k < NG) st_part(dsub + (size_t)rec * RECF + 4 * k, R, min(4, RECF - 4 * k));
      };
      // two register sets used alternately: the next step's LDS reads are in flight while the current one is evaluated
      float4 A0 = wS[0][r16], B0 = wS[1][r16], C0 = wS[2][r16];
      uint32_t t0 = wR[r16];
      for (int s = 0; s < cnt; s += 2) {
        const int s1 = s + 1 < cnt ? s + 1 : s;
        const float4 A1 = wS[0][r16 + s1], B1 = wS[1][r16 + s1], C1 = wS[2][r16 + s1];
        const uint32_t t1 = wR[r16 + s1];
        step(A0, B0, C0, t0, s);
        if (s + 1 < cnt) {
          const int s2 = s + 2 < cnt ? s + 2 : s1;
          A0 = wS[0][r16 + s2]; B0 = wS[1][r16 + s2]; C0 = wS[2][r16 + s2];
          t0 = wR[r16 + s2];
          step(A1, B1, C1, t1, s1);
        }
      }
      stg[wv][cur ^ 1][0][lane] = rn0.A; stg[wv][cur ^ 1][1][lane] = rn0.B; stg[wv][cur ^ 1][2][lane] = rn0.C; srec[wv][cur ^ 1][lane] = en0.y;
      en0 = enn0;
    }
  };
  if (z45_wave) run_chunks(std::true_type{});
  else run_chunks(std::false_type{});
}

void launch_composite_bwd2_slam(const CamDev& cam, bool tracking, GeomView g, ImageView iv, BinView b, size_t N_cap, const float* dL,
                                float* dsub, hipStream_t s, const TrackLoss* tl, int dl_planes, int gen) {
  uint32_t ncap = (uint32_t)(N_cap > 0xffffffffull ? 0xffffffffull : N_cap);
  int T = cam.gx * cam.gy;
  int grid = ((T + 7) / 8) * 8;
  TrackLoss none = {};
  if (gen == 4) {
    if (tracking)
      hipLaunchKernelGGL((composite_bwd4_kernel<2>), dim3(grid), dim3(128), 0, s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
    else
      hipLaunchKernelGGL((composite_bwd4_kernel<1>), dim3(grid), dim3(128), 0, s, cam, g, iv, b, ncap, dL, dsub, 0, none, dl_planes);
    return;
  }
  if (gen == 3) {
    if (tracking)
      hipLaunchKernelGGL((composite_bwd3_kernel<2>), dim3(grid), dim3(256), 0, s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
    else
      hipLaunchKernelGGL((composite_bwd3_kernel<1>), dim3(grid), dim3(256), 0, s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
    return;
  }
  if (tracking)
    hipLaunchKernelGGL((composite_bwd2_kernel<2>), dim3(grid), dim3(128), 0, s, cam, g, iv, b, ncap, dL, dsub, tl ? 1 : 0, tl ? *tl : none, dl_planes);
  else
    hipLaunchKernelGGL((composite_bwd2_kernel<1>), dim3(grid), dim3(128), 0, s, cam, g, iv, b, ncap, dL, dsub, 0, none, dl_planes);
}
