// Internal definitions shared by the gfx950 kernels of libmm3dgs_hip.so.
// Data layout in HBM (all caller-owned):
//   geom_state   : Splat[P] (48 B packed AoS record, gathered by id in the compositor) | depth[P] | rect[P] (2x u32)
//                  | clamped[P] (u8, SH clamp bits) | tileoff[P] (u32, workgroup-local exclusive scan of tiles touched)
//                  | block_tiles[ceil(P/256)+1] (u32, tiles touched per preprocess workgroup -> exclusive prefix)
//   image_state  : Mm3dgsHeader | tile_count[T] | ranges[T+1] | cursor[T] | subcount[16T] | final_T[H*W] | n_contrib[H*W]
//   binning_state: keys[N_cap] (u64 = depth_bits<<32 | id; bins are contiguous per tile; after the sort: the bin in sorted order as
//                  block mask | per-tile record << 32) | sublist[16*N_cap] (uint2 {id, -}: depth-ordered list of each 4x4-pixel block;
//                  block L = 4 * (8x8 sub-tile) + (block in the sub-tile) of a tile with bin [start,end) owns
//                  [16*start + L*len, +subcount[16*tile + L])) | payload[N_cap] (direct bins: block mask | per-tile record << 32 by slot)
//   bwd scratch  : dsub[16*N_cap] (records of up to 12 floats; one per (block, splat), written once by the owning 16-lane row of a
//                  wave -- no atomics), LIST-major since round 6 in every mode: the record of the k-th entry of block list L of a tile
//                  is 16*start + L*len + k, a row's walk writes consecutive records
//                  | dtile[N_cap] (one record per (tile, splat) pair = the sum of the pair's block records, formed by the compositor's
//                  workgroup after its rows are done (composite.hip, per-tile combine); a Gaussian's pairs -- its tile rectangle,
//                  row-major -- are contiguous: packed bins at the Gaussian-major pair index block_tiles[id>>8] + tileoff[id] + k,
//                  direct bins inside the projection workgroup's span)
//                  | campartial[ceil(P/256)][32]
// A wave composites one 8x8 sub-tile; each of its four 16-lane rows owns a 4x4 block and walks that block's own list, so a
// wave iteration evaluates up to four different splats (SLAM splats cover ~40 pixels: with one list per sub-tile 85 % of the
// lanes of an iteration were outside the splat; per-block lists need 0.61x the wave iterations).
#define NLIST 16
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mm3dgs.h"

#define TILE 16
#define TILE_PIX 256
#define SPLAT_F MM3DGS_SPLAT_FLOATS
#define MAX_FUSED_SCAN_TILES 4096  // scatter_scan_kernel keeps 2 T words of LDS
#define MAX_LDS_TILES 12288  // per-workgroup LDS tile histogram (48 KB) covers up to ~1920x1600

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define POSEREC_F 20   // GeomView.poserec: Kp[3][2] | Kq[3][3] | x[3] | 2 unused = five float4
struct GeomView {
  float* splat;       // [P][12]
  float* depth;       // [P]
  uint32_t* rect;     // [P][2]  (minx | miny<<16), (maxx | maxy<<16)
  uint8_t* clamped;   // [P]
  uint32_t* tileoff;  // [P]
  uint32_t* block_tiles;  // [ceil(P/256)+1]
  uint32_t* blkoff;   // [P]  workgroup-local exclusive scan of the 4x4 blocks in each splat's block rectangle
  uint32_t* block_blk;    // [ceil(P/256)+1]  blocks per preprocess workgroup -> exclusive prefix
  float* poserec;         // [P][POSEREC_F] (round 6, tracking): the linear map from a splat's screen-space gradient moments to dL/d(camera-space mean)
                          // and its world position -- written by the projection stage of mm3dgs_slam_track, applied per (block, splat) by the
                          // tracking compositor (composite.hip), so that a tracking iteration writes no gradient records at all
};

static inline size_t geom_bytes_impl(int P) {
  size_t p = (size_t)P;
  return align_up(p * SPLAT_F * 4, 256) + align_up(p * 4, 256) + align_up(p * 8, 256) + align_up(p, 256) +
         2 * (align_up(p * 4, 256) + align_up(((p + 255) / 256 + 1) * 4, 256)) + align_up(p * POSEREC_F * 4, 256);
}
static inline GeomView geom_view(void* base, int P) {
  size_t p = (size_t)P;
  char* c = (char*)base;
  GeomView g;
  g.splat = (float*)c;      c += align_up(p * SPLAT_F * 4, 256);
  g.depth = (float*)c;      c += align_up(p * 4, 256);
  g.rect = (uint32_t*)c;    c += align_up(p * 8, 256);
  g.clamped = (uint8_t*)c;  c += align_up(p, 256);
  g.tileoff = (uint32_t*)c; c += align_up(p * 4, 256);
  g.block_tiles = (uint32_t*)c; c += align_up(((p + 255) / 256 + 1) * 4, 256);
  g.blkoff = (uint32_t*)c;  c += align_up(p * 4, 256);
  g.block_blk = (uint32_t*)c; c += align_up(((p + 255) / 256 + 1) * 4, 256);
  g.poserec = (float*)c;
  return g;
}

struct ImageView {
  Mm3dgsHeader* hdr;
  uint32_t* tile_count;  // [T]
  uint32_t* ranges;      // [T+1]
  uint32_t* cursor;      // [T]
  uint32_t* subcount;    // [16T]
  float* final_T;        // [H*W]
  uint32_t* n_contrib;   // [H*W]
  uint32_t* tile_order;  // [max(8 * ceil(T / 8), 8 * TILE_SPAN_SLOTS)]: workgroup -> tile + 1 (valid while hdr->tile_order_tiles == tile_order_key(H, W)): binning.hip tile_order_kernel
  size_t zero_bytes;     // hdr + tile_count: cleared at the start of every forward
};
static inline int tiles_x(int W) { return (W + TILE - 1) / TILE; }
static inline int tiles_y(int H) { return (H + TILE - 1) / TILE; }
// SLAM compositors with the load-balanced tile table: every XCD owns TILE_SPAN_SLOTS workgroup slots (blockIdx = 8 i + x, i < TILE_SPAN_SLOTS),
// of which its load-cut span of tiles fills the first ones (binning.hip tile_order_kernel)
#define TILE_SPAN_SLOTS 256
static inline size_t tile_order_words(size_t T) { size_t n = ((T + 7) / 8) * 8; return n > 8 * TILE_SPAN_SLOTS ? n : (size_t)8 * TILE_SPAN_SLOTS; }
// workgroup slots per XCD that the table covers: the equal-count spans' ceil(T / 8)
static inline __host__ __device__ int slam_span_slots(int T) { return (T + 7) >> 3; }
#ifdef __HIPCC__
// Inclusive prefix sum over the 64 lanes of a wave on the VALU's own crossbars: four shifted adds inside each 16-lane row (row_shr with
// zero fill), then the last lane of row 0 / row 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31) -- six
// DPP adds instead of six ds_bpermute round trips through the LDS crossbar (26 ns of latency each, tools/ubench) plus their selects.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);     // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);     // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);     // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);     // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
  return x;
}
#endif

static inline size_t image_bytes_impl(int H, int W) {
  size_t T = (size_t)tiles_x(W) * tiles_y(H), px = (size_t)H * W;
  return 256 + align_up(T * 4, 256) + align_up((T + 1) * 4, 256) + align_up(T * 4, 256) + align_up(T * NLIST * 4, 256) +
         align_up(px * 4, 256) + align_up(px * 4, 256) + align_up(tile_order_words(T) * 4, 256);
}
// (the header block is 256 bytes, Mm3dgsHeader its first words)
static inline ImageView image_view(void* base, int H, int W) {
  size_t T = (size_t)tiles_x(W) * tiles_y(H), px = (size_t)H * W;
  char* c = (char*)base;
  ImageView v;
  v.hdr = (Mm3dgsHeader*)c;     c += 256;
  v.tile_count = (uint32_t*)c;  c += align_up(T * 4, 256);
  v.zero_bytes = (size_t)(c - (char*)base);
  v.ranges = (uint32_t*)c;      c += align_up((T + 1) * 4, 256);
  v.cursor = (uint32_t*)c;      c += align_up(T * 4, 256);
  v.subcount = (uint32_t*)c;    c += align_up(T * NLIST * 4, 256);
  v.final_T = (float*)c;        c += align_up(px * 4, 256);
  v.n_contrib = (uint32_t*)c;   c += align_up(px * 4, 256);
  v.tile_order = (uint32_t*)c;
  return v;
}
static_assert(sizeof(Mm3dgsHeader) <= 256, "the header block is 256 bytes");

// what Mm3dgsHeader.tile_order_tiles holds while image_state's workgroup -> tile table is valid: the image size it was built for (the
// table sits behind final_T / n_contrib, so its offset depends on H * W, not on the tile count alone); 0 = no table
static inline __host__ __device__ uint32_t tile_order_key(int H, int W) {
  return (H > 0 && W > 0 && H < 65536 && W < 65536) ? (((uint32_t)H << 16) | (uint32_t)W) : 0u;
}

struct BinView {
  unsigned long long* keys;  // [N_cap]
  uint2* sublist;            // [16*N_cap]
  uint16_t* submask;         // [N_cap]
  unsigned long long* payload;  // [N_cap] by position in the tile's bin (direct bins: slot; packed bins: sorted position): block mask | block-rectangle width << 16 | record of the tile's first block << 32
  uint32_t* trec;               // [N_cap] same indexing: the pair's per-tile gradient record (SLAM modes; ~0u: none)
};
static inline size_t binning_bytes_impl(size_t N) {
  if (N < 1) N = 1;
  return align_up(N * 8, 256) + align_up(N * NLIST * 8, 256) + align_up(N * 2, 256) + align_up(N * 8, 256) + align_up(N * 4, 256);
}
static inline BinView bin_view(void* base, size_t N) {
  if (N < 1) N = 1;
  char* c = (char*)base;
  BinView b;
  b.keys = (unsigned long long*)c;  c += align_up(N * 8, 256);
  b.sublist = (uint2*)c;            c += align_up(N * NLIST * 8, 256);
  b.submask = (uint16_t*)c;         c += align_up(N * 2, 256);
  b.payload = (unsigned long long*)c;  c += align_up(N * 8, 256);
  b.trec = (uint32_t*)c;
  return b;
}

// Direct bins (Mm3dgsHeader.bin_cap != 0): key = depth bits << 32 | Gaussian id << slot_bits | slot in the tile's span
// (ids are unique inside a tile, so the slot never decides the order; it leads the sorted entry back to its payload).
// The split of the low word adapts to the map: slot_bits = 32 - bits(P - 1), at most DIRECT_SLOT_BITS_MAX (13: spans of up to
// 8191 pairs for maps of up to 512 k Gaussians; 12 bits at 1 M, 10 bits -- spans of 1023 -- at 4 M).
#define DIRECT_SLOT_BITS_MAX 13
#define DIRECT_SLOT_BITS_MIN 10
static inline int direct_slot_bits(int P) {
  int idb = 1;
  while (idb < 31 && (1ll << idb) < (long long)P) idb++;
  int sb = 32 - idb;
  return sb > DIRECT_SLOT_BITS_MAX ? DIRECT_SLOT_BITS_MAX : sb;
}

#ifdef __HIPCC__
// bin of a tile: [start, start + len) of keys / payload; its block lists start at sublist + NLIST * start
__device__ __forceinline__ void tile_span(const ImageView& iv, int tile, uint32_t N_cap, uint32_t& start, uint32_t& len) {
  const uint32_t cap = iv.hdr->bin_cap;
  const uint32_t a = iv.ranges[tile], b = iv.ranges[tile + 1];
  if (cap) { start = (uint32_t)tile * cap; len = min(a, cap); }
  else { start = min(a, N_cap); len = min(b, N_cap) - start; }
}
#endif

struct BwdView {
  float* dsub;        // [16*N_cap][12] per-(4x4 block, splat) screen-space gradient records
  float* dtile;       // [N_cap][12] SLAM modes: one record per (tile, splat) pair = the sum of the pair's block records, formed by the
                      // compositor's workgroup after its rows are done; a Gaussian's pair records are contiguous (tile rectangle, row-major).
                      // Directly behind dsub: the kernels find it at dsub + NLIST * N_cap * SPLAT_F
  float* campartial;  // [nrows][32] per-workgroup camera-gradient partial sums (SLAM path: float rows; generic path:
                      // the same region read as double rows -- it is sized for doubles)
  int nrows;
};
static inline int bwd_rows(int P) { return (P + 255) / 256; }
static inline size_t bwd_bytes_impl(int P, size_t N) {
  if (N < 1) N = 1;
  return align_up(N * NLIST * SPLAT_F * 4, 256) + align_up(N * SPLAT_F * 4, 256) + align_up((size_t)(bwd_rows(P) + 1) * 32 * 8, 256);
}
static inline BwdView bwd_view(void* base, int P, size_t N) {
  if (N < 1) N = 1;
  char* c = (char*)base;
  BwdView b;
  b.dsub = (float*)c;  c += align_up(N * NLIST * SPLAT_F * 4, 256);
  b.dtile = (float*)c; c += align_up(N * SPLAT_F * 4, 256);
  b.campartial = (float*)c;
  b.nrows = bwd_rows(P);
  return b;
}

// Block rectangle of a splat: the 4x4-pixel blocks (global block coordinates) that its { alpha >= 1/255 } bound can reach,
// clipped to its tile rectangle.  A superset of the blocks sort_tiles_kernel lists (same bound, 0.01 px of slack against
// the different rounding of tile-relative arithmetic).  Used by the projection kernels (size), the sort (record index
// of every list entry) and the backward projection (gather).  A, B = first 32 bytes of the splat record.
struct BlkRect { int bx0, by0, bw, bh; };
__device__ __forceinline__ BlkRect block_rect(const float4 A, const float4 B, uint32_t r0, uint32_t r1) {
  const int X0 = (int)(r0 & 0xffff) * 4, Y0 = (int)(r0 >> 16) * 4, X1 = (int)(r1 & 0xffff) * 4, Y1 = (int)(r1 >> 16) * 4;
  BlkRect q;
  q.bx0 = X0; q.by0 = Y0; q.bw = X1 - X0; q.bh = Y1 - Y0;
  if (q.bw <= 0 || q.bh <= 0) { q.bw = 0; q.bh = 0; return q; }
  // (rounded intrinsics: the rectangle is recomputed in several kernels and must come out identical in all of them)
  const float tau = __logf(__fmul_rn(255.f, B.y));
  const float det = __fsub_rn(__fmul_rn(A.z, B.x), __fmul_rn(A.w, A.w));
  if (!(det > 0.f)) return q;                      // degenerate conic: no culling
  if (!(tau > 0.f)) { q.bw = 0; q.bh = 0; return q; }
  const float k = __fdiv_rn(__fmul_rn(2.f, tau), det);
  const float hx = __fadd_rn(__fmul_rn(__fsqrt_rn(__fmul_rn(k, B.x)), 1.0002f), 0.012f);
  const float hy = __fadd_rn(__fmul_rn(__fsqrt_rn(__fmul_rn(k, A.z)), 1.0002f), 0.012f);
  // block b covers pixel centres [4b, 4b+3]:  overlap  <=>  c - h <= 4b + 3  and  c + h >= 4b
  const int bx0 = max(X0, (int)ceilf(__fmul_rn(__fsub_rn(__fsub_rn(A.x, hx), 3.f), 0.25f))), bx1 = min(X1 - 1, (int)floorf(__fmul_rn(__fadd_rn(A.x, hx), 0.25f)));
  const int by0 = max(Y0, (int)ceilf(__fmul_rn(__fsub_rn(__fsub_rn(A.y, hy), 3.f), 0.25f))), by1 = min(Y1 - 1, (int)floorf(__fmul_rn(__fadd_rn(A.y, hy), 0.25f)));
  q.bx0 = bx0; q.by0 = by0; q.bw = max(bx1 - bx0 + 1, 0); q.bh = max(by1 - by0 + 1, 0);
  if (q.bw == 0 || q.bh == 0) { q.bw = 0; q.bh = 0; }
  return q;
}

// Camera constants copied by value into kernel arguments (matrices stay device pointers: they are torch
// tensors computed differentiably on the device, slam/renderer.py:117-124, and must not cost a host sync).
struct CamDev {
  int H, W, gx, gy;
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int sh_degree;
  int tilemap;  // 0: tile = workgroup id; 1 (default): contiguous tile span per XCD
  int tile_table; // SLAM entry points with persistent state: honour image_state's load-balanced workgroup -> tile table when it is valid
  int stats;    // count diagnostics into the header (MM3DGS_STATS=1)
#ifdef MM3DGS_PROBES
  int probes;   // MM3DGS_EXP bits (see PROBE below)
#endif
  int sort_single;  // 1: a single sort launch (16 KB LDS tier + global-memory path for longer lists)
  int bg_extras;    // 1 (SLAM entry points): channels 3..5 are the depth bundle of a second reference pass and get T_final * bg[ch - 3] as well
  uint32_t trec_cap; // direct bins: per-tile gradient records per projection workgroup (workgroup w owns [w * trec_cap, (w + 1) * trec_cap)); 0: packed bins (Gaussian-major pair index)
  int state_clean;  // 1: persistent state buffers (MM3DGS_FWD_STATE_CLEAN): every forward leaves tile_count[] and cursor[] zero for the next one
  int fused_scan;   // 1: no scan_tiles launch, every scatter workgroup scans the tile counters itself (persistent state)
  const float* bg;
  const float* view;
  const float* proj;
  const float* campos;
};
// Timing probes of the hot kernels (tools/late_probe.sh, tools/exp_ab.sh: "how long does the launch take without phase X" -- the results of such a
// run are INVALID).  They exist only in a library built with -DMM3DGS_PROBES (tools/build_variant.sh), where the environment variable MM3DGS_EXP
// selects them by bit; in the product build PROBE() is the constant 0, the probed branches fold away and the variable is never read.
//   bit 0: backward compositor without its record stores | 1: sort phase only | 2: no list emission | 3: no sort | 4: combine without its
//   record gather | 5: no combine | 6: binning without the shared big-splat list | 7: only the own pairs counted | 8: no > 32-tile counting
//   9: backward compositor without its main loop | 10: mapping backward compositor without the loss prologue | 11: combine without its stores
#ifdef MM3DGS_PROBES
#define PROBE(cam, bit) ((((cam).probes) >> (bit)) & 1)
#define PROBE_WORD(cam) ((cam).probes)
#else
#define PROBE(cam, bit) 0
#define PROBE_WORD(cam) 0
#endif
static inline int env_flag(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static inline CamDev cam_dev(const Mm3dgsCamera* c) {
  CamDev d;
  d.H = c->image_height; d.W = c->image_width;
  d.gx = tiles_x(d.W); d.gy = tiles_y(d.H);
  d.tanfovx = c->tanfovx; d.tanfovy = c->tanfovy;
  d.focal_x = d.W / (2.0f * c->tanfovx);
  d.focal_y = d.H / (2.0f * c->tanfovy);
  d.scale_modifier = c->scale_modifier;
  d.sh_degree = c->sh_degree;
  d.tilemap = env_flag("MM3DGS_TILEMAP", 1);   // contiguous tile span per XCD: the backward compositor measured 4 % faster (splat records and lists stay in one L2)
  d.stats = env_flag("MM3DGS_STATS", 0);
#ifdef MM3DGS_PROBES
  d.probes = env_flag("MM3DGS_EXP", 0);
#endif
  d.sort_single = 0;
  d.bg_extras = 0;
  d.state_clean = 0;
  d.tile_table = 0;
  d.trec_cap = 0;
  d.fused_scan = 0;
  d.bg = c->bg; d.view = c->viewmatrix; d.proj = c->projmatrix; d.campos = c->campos;
  return d;
}

// ---- launchers implemented in the individual .hip files ----------------------------------------------------
void launch_preprocess_fwd(const CamDev& cam, int P, int M, int C, const float* means3D, const float* shs,
                           const float* colors, const float* opac, const float* scales, const float* rots,
                           const float* cov3d, int32_t* radii, GeomView g, ImageView iv, hipStream_t s);
void launch_scan_tiles(int T, int P, GeomView g, ImageView iv, hipStream_t s, int sticky = 0);
bool launch_tile_order(int T, int H, int W, ImageView iv, hipStream_t s);   // load-balanced workgroup -> tile table of the SLAM compositors (binning.hip)
void launch_scatter_sort(const CamDev& cam, int P, GeomView g, ImageView iv, BinView b, size_t N_cap,
                         const int32_t* radii_or_null, hipStream_t s, bool scatter_only = false);
void launch_composite_fwd(const CamDev& cam, int C, GeomView g, ImageView iv, BinView b, size_t N_cap,
                          float* out_color, hipStream_t s);
void launch_composite_bwd(const CamDev& cam, int C, GeomView g, ImageView iv, BinView b, size_t N_cap,
                          const float* dL_dout, float* dsub, hipStream_t s);
void launch_preprocess_bwd(const CamDev& cam, int P, int M, int C, const float* means3D, const float* shs,
                           const float* colors, const float* opac, const float* scales, const float* rots,
                           const float* cov3d, const int32_t* radii, GeomView g, BinView b, size_t N_cap, BwdView bw,
                           float* dmeans3D,
                           float* dmeans2D, float* dshs, float* dcolors, float* dopac, float* dscales,
                           float* drots, float* dcov3d, bool want_cam, int flags, hipStream_t s);
void launch_camgrad_finish(BwdView bw, float* dview, float* dproj, float* dcampos, hipStream_t s);
void launch_mark_visible(const CamDev& cam, int P, const float* means3D, uint8_t* visible, hipStream_t s);

#ifdef __HIPCC__
// ---- wave64 helpers ------------------------------------------------------------------------------------------
// Sum across the 64 lanes with DPP row shifts + row broadcasts (6 VALU adds, no LDS); total lands in lane 63.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_f<0x111, 0xf, 0xf>(v);  // row_shr:1
  v += dpp_f<0x112, 0xf, 0xf>(v);  // row_shr:2
  v += dpp_f<0x114, 0xf, 0xf>(v);  // row_shr:4
  v += dpp_f<0x118, 0xf, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row sum
  v += dpp_f<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1,3
  v += dpp_f<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2,3
  return v;
}
// the same reduction on doubles (DPP on the two 32-bit halves + v_add_f64): for sums that cancel (camera gradients, Pearson
// moments), where a float reduction's 1e-7 relative error would be amplified by |terms| / |total|
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, BANK_MASK, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_to_lane63_f64(double v) {
  v += dpp_d<0x111, 0xf, 0xf>(v);
  v += dpp_d<0x112, 0xf, 0xf>(v);
  v += dpp_d<0x114, 0xf, 0xf>(v);
  v += dpp_d<0x118, 0xf, 0xf>(v);
  v += dpp_d<0x142, 0xa, 0xf>(v);
  v += dpp_d<0x143, 0xc, 0xf>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = wave_sum_to_lane63(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif
