// C-ABI entry points of libmm3dgs_hip.so (declared in include/mm3dgs.h).  Plain pointers and sizes only; every
// launch goes to the caller's stream; no device allocation, no host synchronisation.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "mm3dgs_common.h"
#include <algorithm>
#include "fused_api.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-10, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

// ---- optional per-kernel timing with HIP events on the caller's stream (bench.py's roofline leg) -------------------
#include <mutex>
#include <vector>
struct KernelProfile {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
  unsigned long long launches = 0;
  double total_ms = 0.0;
};
static KernelProfile g_prof[MM3DGS_PROF_KERNELS];
static std::mutex g_prof_mu;
static int g_prof_on = 0;
struct ProfScope {
  int k; hipStream_t s; hipEvent_t e1 = nullptr; bool on;
  // g_prof_on: 0 off, 1 every kernel, 2 only the compositors (the roofline kernels) and only every 64th launch of
  // it: an event pair around EVERY launch costs ~4 % of the SLAM frame rate (measured), a 1-in-16 sample 0.7 % (measured, round 3), 1-in-64 0.2 %
  ProfScope(int k_, hipStream_t s_) : k(k_), s(s_), on(g_prof_on == 1 || (g_prof_on == 2 && (k_ == MM3DGS_PROF_COMPOSITE_BWD || k_ == MM3DGS_PROF_COMPOSITE_BWD_TRACK ||
                                                                                     k_ == MM3DGS_PROF_COMPOSITE_FWD || k_ == MM3DGS_PROF_TRACK_FWD_BWD))) {
    if (on && g_prof_on == 2) {
      static unsigned long long sample[MM3DGS_PROF_KERNELS] = {};
      on = (sample[k_]++ & 63ull) == 0ull;
    }
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!g_prof[k].pool.empty()) { ev = g_prof[k].pool.back(); g_prof[k].pool.pop_back(); }
    else { (void)hipEventCreate(&ev.first); (void)hipEventCreate(&ev.second); }
    g_prof[k].pending.push_back(ev);
    e1 = ev.second;
    (void)hipEventRecord(ev.first, s);
  }
  ~ProfScope() { if (on) (void)hipEventRecord(e1, s); }
};

__global__ void profile_null_kernel() {}

extern "C" {

void mm3dgs_profile_enable(int on) { g_prof_on = on; }
// What an event pair adds to the interval it brackets: T1 = (event, kernel, event), T2 = (event, kernel, kernel, event) with an empty
// kernel; T2 - T1 is what one more launch costs, so 2 T1 - T2 is the part of T1 that is not the launch.  Synchronises the stream.
double mm3dgs_profile_event_overhead_ms(void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0.0;
  double t[2] = {0.0, 0.0};
  const int reps = 32;
  for (int n = 1; n <= 2; n++) {
    for (int r = 0; r < reps + 4; r++) {
      (void)hipEventRecord(a, s);
      for (int q = 0; q < n; q++) hipLaunchKernelGGL(profile_null_kernel, dim3(1), dim3(64), 0, s);
      (void)hipEventRecord(b, s);
      (void)hipEventSynchronize(b);
      float ms = 0.f;
      if (r >= 4 && hipEventElapsedTime(&ms, a, b) == hipSuccess) t[n - 1] += ms;
    }
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  const double o = (2.0 * t[0] - t[1]) / reps;
  return o > 0.0 ? o : 0.0;
}
int mm3dgs_profile_read(int kernel, uint64_t* launches, double* total_ms) {
  if (kernel < 0 || kernel >= MM3DGS_PROF_KERNELS) return fail(-1, "bad kernel id %d", kernel);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  KernelProfile& p = g_prof[kernel];
  for (auto& ev : p.pending) {
    if (hipEventSynchronize(ev.second) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { p.total_ms += ms; p.launches++; }
    }
    p.pool.push_back(ev);
  }
  p.pending.clear();
  if (launches) *launches = p.launches;
  if (total_ms) *total_ms = p.total_ms;
  p.launches = 0; p.total_ms = 0.0;
  return 0;
}

const char* mm3dgs_last_error(void) { return g_err; }
int mm3dgs_version(void) { return MM3DGS_ABI_VERSION; }   // (history in include/mm3dgs.h)

size_t mm3dgs_geom_bytes(int P) { return geom_bytes_impl(P > 0 ? P : 1); }
size_t mm3dgs_image_bytes(int H, int W) { return image_bytes_impl(H, W); }
size_t mm3dgs_binning_bytes(size_t N) { return binning_bytes_impl(N); }
size_t mm3dgs_backward_scratch_bytes(int P, size_t N) { return bwd_bytes_impl(P, N); }

static int check_common(const Mm3dgsCamera* cam, int P, int M, int C, const float* shs, const float* colors,
                        const float* scales, const float* rots, const float* cov3d) {
  if (!cam) return fail(-1, "camera is NULL");
  if (cam->image_height <= 0 || cam->image_width <= 0) return fail(-1, "bad image size %dx%d", cam->image_height, cam->image_width);
  if (cam->image_height > 16 * 65535 || cam->image_width > 16 * 65535) return fail(-1, "image too large for 16-bit tile coordinates");
  if (P < 0) return fail(-1, "P < 0");
  if (C < 1 || C > MM3DGS_MAX_CHANNELS) return fail(-1, "C=%d outside 1..%d", C, MM3DGS_MAX_CHANNELS);
  if (P == 0) return 0;  // nothing to validate: empty tensors arrive as NULL pointers
  if (!shs && !colors) return fail(-2, "Please provide excatly one of either SHs or precomputed colors!");
  if (shs) {
    if (C < 3) return fail(-2, "SH colour needs C >= 3");
    if (C > 3 && !colors) return fail(-2, "C=%d with SHs needs %d extra precomputed channels", C, C - 3);
    if (cam->sh_degree < 0 || cam->sh_degree > 3) return fail(-2, "sh_degree %d outside 0..3", cam->sh_degree);
    if (M < (cam->sh_degree + 1) * (cam->sh_degree + 1)) return fail(-2, "M=%d too small for sh_degree %d", M, cam->sh_degree);
  }
  if ((scales == nullptr || rots == nullptr) == (cov3d == nullptr))
    return fail(-2, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  if (!cam->bg || !cam->viewmatrix || !cam->projmatrix || !cam->campos) return fail(-1, "camera device pointers missing");
  return 0;
}

int mm3dgs_forward_geom(const Mm3dgsCamera* cam, int P, int M, int C, const float* means3D, const float* shs,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_state,
                        void* image_state, uint32_t* host_num_rendered, void* stream) {
  int rc = check_common(cam, P, M, C, shs, colors_precomp, scales, rotations, cov3D_precomp);
  if (rc) return rc;
  if (!geom_state || !image_state || (P > 0 && (!means3D || !opacities || !radii))) return fail(-1, "NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  CamDev cd = cam_dev(cam);
  GeomView g = geom_view(geom_state, P > 0 ? P : 1);
  ImageView iv = image_view(image_state, cd.H, cd.W);
  if (hipMemsetAsync(image_state, 0, iv.zero_bytes, s) != hipSuccess) return fail(-10, "memset failed");
  { ProfScope ps(MM3DGS_PROF_PREPROCESS_FWD, s);
    launch_preprocess_fwd(cd, P, M, C, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, g, iv, s); }
  { ProfScope ps(MM3DGS_PROF_SCAN, s); launch_scan_tiles(cd.gx * cd.gy, P, g, iv, s); }
  if (host_num_rendered)
    if (hipMemcpyAsync(host_num_rendered, &iv.hdr->num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess)
      return fail(-10, "num_rendered copy failed");
  return check_launch("forward_geom");
}

int mm3dgs_forward_raster(const Mm3dgsCamera* cam, int P, int C, const void* geom_state, void* image_state,
                          void* binning_state, size_t N_capacity, float* out_color, void* stream) {
  if (!cam || !geom_state || !image_state || !binning_state || !out_color) return fail(-1, "NULL buffer");
  if (C < 1 || C > MM3DGS_MAX_CHANNELS) return fail(-1, "C=%d outside 1..%d", C, MM3DGS_MAX_CHANNELS);
  hipStream_t s = (hipStream_t)stream;
  CamDev cd = cam_dev(cam);
  GeomView g = geom_view((void*)geom_state, P > 0 ? P : 1);
  ImageView iv = image_view(image_state, cd.H, cd.W);
  BinView b = bin_view(binning_state, N_capacity);
  { ProfScope ps(MM3DGS_PROF_BIN_SORT, s); launch_scatter_sort(cd, P, g, iv, b, N_capacity, nullptr, s); }
  { ProfScope ps(MM3DGS_PROF_COMPOSITE_FWD, s); launch_composite_fwd(cd, C, g, iv, b, N_capacity, out_color, s); }
  return check_launch("forward_raster");
}

int mm3dgs_forward(const Mm3dgsCamera* cam, int P, int M, int C, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales,
                   const float* rotations, const float* cov3D_precomp, float* out_color, int32_t* radii,
                   void* geom_state, void* image_state, void* binning_state, size_t N_capacity, void* stream) {
  int rc = mm3dgs_forward_geom(cam, P, M, C, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                               radii, geom_state, image_state, nullptr, stream);
  if (rc) return rc;
  return mm3dgs_forward_raster(cam, P, C, geom_state, image_state, binning_state, N_capacity, out_color, stream);
}

int mm3dgs_backward(const Mm3dgsCamera* cam, int P, int M, int C, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* opacities, const float* scales,
                    const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                    const void* geom_state, const void* image_state, const void* binning_state,
                    size_t N_capacity, const float* dL_dout, void* backward_scratch, float* dL_dmeans3D,
                    float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors_precomp, float* dL_dopacities,
                    float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dview, float* dL_dproj,
                    float* dL_dcampos, int flags, void* stream) {
  int rc = check_common(cam, P, M, C, shs, colors_precomp, scales, rotations, cov3D_precomp);
  if (rc) return rc;
  if (!geom_state || !image_state || !binning_state || !dL_dout || !backward_scratch || (P > 0 && !radii))
    return fail(-1, "NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  CamDev cd = cam_dev(cam);
  GeomView g = geom_view((void*)geom_state, P > 0 ? P : 1);
  ImageView iv = image_view((void*)image_state, cd.H, cd.W);
  BinView b = bin_view((void*)binning_state, N_capacity);
  BwdView bw = bwd_view(backward_scratch, P, N_capacity);
  { ProfScope ps(MM3DGS_PROF_COMPOSITE_BWD, s); launch_composite_bwd(cd, C, g, iv, b, N_capacity, dL_dout, bw.dsub, s); }
  ProfScope ps_pb(MM3DGS_PROF_PREPROCESS_BWD, s);
  bool want_cam = dL_dview || dL_dproj || dL_dcampos;
  launch_preprocess_bwd(cd, P, M, C, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, g, b,
                        N_capacity, bw, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors_precomp, dL_dopacities, dL_dscales, dL_drotations,
                        dL_dcov3D, want_cam, flags, s);
  if (want_cam) launch_camgrad_finish(bw, dL_dview, dL_dproj, dL_dcampos, s);
  return check_launch("backward");
}

// ---- developer switches (environment) ---------------------------------------------------------------------------------------
// Every switch the library reads is listed in include/mm3dgs.h ("Environment").  They exist for the bit-identity / equivalence tests and for same-box
// A/Bs; one left set in a user's environment silently changes which kernels run -- so the first SLAM entry point of a process says which are set.
static void warn_env_once() {
  static const bool done = [] {
    static const char* const names[] = {"MM3DGS_NO_DIRECT_BINS", "MM3DGS_NO_FUSED_SORT", "MM3DGS_NO_FUSED_SCAN", "MM3DGS_NO_FUSED_TRACK", "MM3DGS_NO_FOLDED_LOSS",
                                        "MM3DGS_NO_FORWARD_ROWS", "MM3DGS_NO_FUSED_PROJECT", "MM3DGS_NO_TILE_ORDER", "MM3DGS_NO_POSE_CHAIN", "MM3DGS_TILEMAP",
                                        "MM3DGS_DIRECT_MAX_TILES", "MM3DGS_STATS", "MM3DGS_SLAM_LDS_PAD", "MM3DGS_FWD_LDS_PAD", "MM3DGS_BWD_LDS_PAD"};
    for (const char* n : names) {
      const char* v = getenv(n);
      if (v && *v) fprintf(stderr, "mm3dgs: developer switch %s=%s is set in the environment: it changes which kernels run (include/mm3dgs.h, \"Environment\")\n", n, v);
    }
    return true;
  }();
  (void)done;
}

// ---- fused SLAM iteration ----------------------------------------------------------------------------------------
static SlamIn slam_in(const Mm3dgsSlamInputs* in) {
  SlamIn s;
  s.pose = in->pose; s.xyz = in->xyz; s.f_dc = in->f_dc; s.opacity = in->opacity; s.scaling = in->scaling;
  s.rotation = in->rotation; s.isotropic = in->isotropic; s.world = in->world_means;
  s.sh_deg = (in->f_rest && in->sh_degree > 0) ? in->sh_degree : 0;
  s.f_rest = s.sh_deg ? in->f_rest : nullptr; s.n_rest = s.sh_deg ? in->n_rest : 0;
  return s;
}
static int check_slam(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in) {
  warn_env_once();
  if (!cam || !in) return fail(-1, "NULL argument");
  if (cam->image_height <= 0 || cam->image_width <= 0) return fail(-1, "bad image size");
  if (P < 0) return fail(-1, "P < 0");
  if (!cam->bg || !cam->projmatrix) return fail(-1, "camera device pointers missing");
  if (!in->pose) return fail(-1, "pose is NULL");
  if (P > 0 && (!in->xyz || !in->f_dc || !in->opacity || !in->scaling || !in->rotation)) return fail(-1, "NULL Gaussian parameter");
  if (in->sh_degree < 0 || in->sh_degree > 3) return fail(-2, "sh_degree %d outside 0..3", in->sh_degree);
  if (in->sh_degree > 0) {
    if (P > 0 && !in->f_rest) return fail(-2, "sh_degree %d needs the f_rest rows", in->sh_degree);
    if (in->n_rest < (in->sh_degree + 1) * (in->sh_degree + 1) - 1 || in->n_rest > 15) return fail(-2, "n_rest = %d does not hold sh_degree %d (or exceeds 15)", in->n_rest, in->sh_degree);
    if (in->world_means) return fail(-2, "an active SH degree > 0 is native in the transform_means_python mode only");
  }
  return 0;
}

static bool slam_fused_sort(int flags) {
  static const int no_fused_sort = env_flag("MM3DGS_NO_FUSED_SORT", 0);
  return (flags & MM3DGS_FWD_SHORT_LISTS) && !no_fused_sort;
}

// direct bins (MM3DGS_FWD_DIRECT_BINS): one decision for the forward and the backward of a render
struct DirectBins { bool on; uint32_t bin_cap, rec_cap, trec_cap; int nblocks, slot_bits; };
static DirectBins slam_direct_bins(int flags, const CamDev& cd, int P, size_t N_capacity) {
  static const int no_direct = env_flag("MM3DGS_NO_DIRECT_BINS", 0);
  static const int no_fused_scan = env_flag("MM3DGS_NO_FUSED_SCAN", 0);
  static const int max_tiles = env_flag("MM3DGS_DIRECT_MAX_TILES", 11264);   // the binning kernel keeps one LDS word per tile beside 18 KB of static LDS: 62 KB at this limit (1920x1080 = 8160 tiles, 1920x1440 = 10800)
  DirectBins d;
  const int T = cd.gx * cd.gy;
  d.nblocks = (P + 255) / 256;
  const size_t nb = (size_t)std::max(d.nblocks, 1);
  d.slot_bits = direct_slot_bits(P);
  d.bin_cap = (uint32_t)std::min<size_t>(N_capacity / (size_t)std::max(T, 1), (size_t)((1u << std::max(d.slot_bits, 1)) - 1u));
  // records of the backward scratch per projection workgroup (the scratch holds NLIST records per pair of capacity)
  d.rec_cap = (uint32_t)std::min<size_t>((size_t)NLIST * N_capacity / nb, 0xffffffffull / nb);
  // per-tile records (one per pair) per projection workgroup: the region holds N_capacity of them
  d.trec_cap = (uint32_t)std::min<size_t>(N_capacity / nb, 0xffffffffull / nb);
  d.on = (flags & MM3DGS_FWD_DIRECT_BINS) && (flags & MM3DGS_FWD_STATE_CLEAN) && slam_fused_sort(flags) && !no_direct && !no_fused_scan &&
         P > 0 && d.slot_bits >= DIRECT_SLOT_BITS_MIN && T <= max_tiles && T <= MAX_LDS_TILES && d.bin_cap >= 32 && d.rec_cap >= 1024 && d.trec_cap >= 256 &&
         N_capacity >= 4 * (size_t)P;
  return d;
}

// The SLAM loops refresh image_state's load-balanced workgroup -> tile table (binning.hip) once per call, from the list lengths of the
// last render; the compositors honour it while the header says it matches the grid (persistent state only: a per-call memset clears it).
static bool slam_tile_table(int flags) { return (flags & MM3DGS_FWD_STATE_CLEAN) && !env_flag("MM3DGS_NO_TILE_ORDER", 0); }
static void slam_refresh_tile_order(const Mm3dgsCamera* cam, void* image_state, int flags, void* stream) {
  if (!cam || !image_state || !slam_tile_table(flags) || cam->image_height <= 0 || cam->image_width <= 0) return;
  if (flags & MM3DGS_FWD_KEEP_TILE_ORDER) return;      // (a table left by an earlier call is honoured while its key matches the image size)
  const CamDev cd = cam_dev(cam);
  launch_tile_order(cd.gx * cd.gy, cd.H, cd.W, image_view(image_state, cd.H, cd.W), (hipStream_t)stream);
}

static int slam_forward_impl(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, float* out_color, int32_t* radii,
                             void* geom_state, void* image_state, void* binning_state, size_t N_capacity, int flags, void* stream,
                             const TrackLoss* tl, float* track_dsub = nullptr, bool projected = false, bool pose_chain = false) {
  int rc = check_slam(cam, P, in);
  if (rc) return rc;
  if (!out_color || !geom_state || !image_state || !binning_state || (P > 0 && !radii)) return fail(-1, "NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  CamDev cd = cam_dev(cam);
  GeomView g = geom_view(geom_state, P > 0 ? P : 1);
  ImageView iv = image_view(image_state, cd.H, cd.W);
  BinView b = bin_view(binning_state, N_capacity);
  if (!(flags & MM3DGS_FWD_STATE_CLEAN))
    if (hipMemsetAsync(image_state, 0, iv.zero_bytes, s) != hipSuccess) return fail(-10, "memset failed");
  cd.sort_single = (flags & MM3DGS_FWD_SHORT_LISTS) ? 1 : 0;
  cd.bg_extras = 1;
  cd.state_clean = (flags & MM3DGS_FWD_STATE_CLEAN) ? 1 : 0;
  cd.tile_table = slam_tile_table(flags) ? 1 : 0;
  // persistent clean state + a tile grid that fits two LDS words per tile: fold the scan into the scatter workgroups
  cd.fused_scan = ((flags & MM3DGS_FWD_STATE_CLEAN) && P > 0 && cd.gx * cd.gy <= MAX_FUSED_SCAN_TILES && !env_flag("MM3DGS_NO_FUSED_SCAN", 0)) ? 1 : 0;
  // short lists (the SLAM regime): the per-tile sort runs inside the forward compositing launch
  const bool fused_sort = slam_fused_sort(flags);
  if (tl && !fused_sort) return fail(-1, "internal: folded tracking loss needs the fused sort path");
  // direct bins: the host sized the binning state as T x (per-tile capacity), so projection and binning are one launch
  const DirectBins db = slam_direct_bins(flags, cd, P, N_capacity);
  cd.trec_cap = db.on ? db.trec_cap : 0u;
  if (db.on) {
    // projected: the previous mapping iteration's backward launch already projected and binned this view (slam_bwd_project_kernel)
    if (!projected) { ProfScope ps(MM3DGS_PROF_PREPROCESS_FWD, s); launch_slam_project_bin(cd, P, slam_in(in), radii, g, iv, b, db.bin_cap, db.rec_cap, db.slot_bits, s, pose_chain); }
    { ProfScope ps(track_dsub ? MM3DGS_PROF_TRACK_FWD_BWD : MM3DGS_PROF_COMPOSITE_FWD, s);
      if (track_dsub) launch_sort_composite_fwd_bwd_track(cd, g, iv, b, N_capacity, out_color, 1, s, *tl, db.nblocks, track_dsub, db.bin_cap, db.slot_bits, pose_chain);
      else launch_sort_composite_fwd6(cd, g, iv, b, N_capacity, out_color, 1, s, tl, db.nblocks, db.bin_cap, db.slot_bits); }
    return check_launch("slam_forward");
  }
  { ProfScope ps(MM3DGS_PROF_PREPROCESS_FWD, s); launch_slam_preprocess_fwd(cd, P, slam_in(in), radii, g, iv, s, nullptr, false, pose_chain); }
  if (!cd.fused_scan) { ProfScope ps(MM3DGS_PROF_SCAN, s); launch_scan_tiles(cd.gx * cd.gy, P, g, iv, s, (flags & MM3DGS_FWD_STATE_CLEAN) ? 1 : 0); }
  { ProfScope ps(MM3DGS_PROF_BIN_SORT, s); launch_scatter_sort(cd, P, g, iv, b, N_capacity, nullptr, s, fused_sort); }
  { ProfScope ps((fused_sort && track_dsub) ? MM3DGS_PROF_TRACK_FWD_BWD : MM3DGS_PROF_COMPOSITE_FWD, s);
    if (fused_sort && track_dsub) launch_sort_composite_fwd_bwd_track(cd, g, iv, b, N_capacity, out_color, (cd.fused_scan || cd.state_clean) ? 1 : 0, s, *tl, 0, track_dsub, 0, DIRECT_SLOT_BITS_MAX, pose_chain);
    else if (fused_sort) launch_sort_composite_fwd6(cd, g, iv, b, N_capacity, out_color, (cd.fused_scan || cd.state_clean) ? 1 : 0, s, tl);
    else launch_composite_fwd(cd, 6, g, iv, b, N_capacity, out_color, s); }
  return check_launch("slam_forward");
}

int mm3dgs_slam_visibility(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, int32_t* radii, uint32_t* seen_count, void* geom_state,
                           void* stream) {
  int rc = check_slam(cam, P, in);
  if (rc) return rc;
  if (!geom_state || (P > 0 && !radii)) return fail(-1, "NULL buffer");
  CamDev cd = cam_dev(cam);
  GeomView g = geom_view(geom_state, P > 0 ? P : 1);
  ImageView iv = {};
  launch_slam_preprocess_fwd(cd, P, slam_in(in), radii, g, iv, (hipStream_t)stream, seen_count, true);
  return check_launch("slam_visibility");
}

int mm3dgs_slam_forward(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, float* out_color, int32_t* radii,
                        void* geom_state, void* image_state, void* binning_state, size_t N_capacity, int flags, void* stream) {
  return slam_forward_impl(cam, P, in, out_color, radii, geom_state, image_state, binning_state, N_capacity, flags, stream, nullptr, nullptr,
                           (flags & MM3DGS_FWD_PROJECTED) != 0);
}

// Mm3dgsMapAdam -> the kernels' scalars: formed in double, rounded once to float (torch.optim.Adam does this arithmetic on Python floats)
static int map_adam_dev(const Mm3dgsMapAdam* map_adam, MapAdam& ma) {
  if (map_adam->step < 1) return fail(-1, "map Adam step must be >= 1");
  for (int i = 0; i < 5; i++) {
    if (!map_adam->param[i] || !map_adam->exp_avg[i] || !map_adam->exp_avg_sq[i]) return fail(-2, "map Adam group %d has a NULL pointer", i);
    ma.p[i] = map_adam->param[i]; ma.m[i] = map_adam->exp_avg[i]; ma.v[i] = map_adam->exp_avg_sq[i];
    ma.step_size[i] = (float)(map_adam->lr[i] / (1.0 - pow(map_adam->beta1, (double)map_adam->step)));
  }
  ma.omb1 = (float)(1.0 - map_adam->beta1); ma.beta2 = (float)map_adam->beta2; ma.omb2 = (float)(1.0 - map_adam->beta2);
  ma.eps = (float)map_adam->eps;
  ma.bc2s = (float)sqrt(1.0 - pow(map_adam->beta2, (double)map_adam->step));
  ma.opt_mask = map_adam->opt_mask;
  ma.rp = map_adam->rest_param; ma.rm = map_adam->rest_exp_avg; ma.rv = map_adam->rest_exp_avg_sq;
  if (ma.rp && (!ma.rm || !ma.rv)) return fail(-2, "map Adam: the f_rest group has a NULL moment pointer");
  ma.rest_step_size = (float)(map_adam->rest_lr / (1.0 - pow(map_adam->beta1, (double)map_adam->step)));
  ma.on = 1;
  return 0;
}

static int slam_backward_impl(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, const int32_t* radii,
                              const void* geom_state, const void* image_state, const void* binning_state, size_t N_capacity,
                              const float* dL_dout, void* backward_scratch, const Mm3dgsSlamGrads* grads, float* dL_dpose,
                              const Mm3dgsPoseAdam* pose_adam, const Mm3dgsMapAdam* map_adam, int flags, void* stream, const TrackLoss* tl,
                              float* prior_loss4 = nullptr, int dl_planes = 6, bool compositor_done = false, const float* fuse_next_pose = nullptr,
                              bool* fused_out = nullptr, bool pose_chain = false) {
  PoseLossScale pls = {nullptr, 0, 0.f, nullptr};
  if (tl && tl->defer_scale) { pls.rows = tl->partial; pls.nrows = ((tl->cfg.W + 15) / 16) * ((tl->cfg.H + 15) / 16); pls.w_l1 = tl->cfg.w_l1; pls.loss4 = tl->loss4; }
  int rc = check_slam(cam, P, in);
  if (rc) return rc;
  if (!geom_state || !image_state || !binning_state || !dL_dout || !backward_scratch || (P > 0 && !radii)) return fail(-1, "NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  CamDev cd = cam_dev(cam);
  GeomView g = geom_view((void*)geom_state, P > 0 ? P : 1);
  ImageView iv = image_view((void*)image_state, cd.H, cd.W);
  BinView b = bin_view((void*)binning_state, N_capacity);
  BwdView bw = bwd_view(backward_scratch, P, N_capacity);
  cd.bg_extras = 1;
  cd.tile_table = slam_tile_table(flags) ? 1 : 0;
  const DirectBins db_bwd = slam_direct_bins(flags, cd, P, N_capacity);
  cd.trec_cap = db_bwd.on ? db_bwd.trec_cap : 0u;
  SlamGrads sg = {};
  if (grads) {
    const bool any = grads->d_xyz || grads->d_f_dc || grads->d_opacity || grads->d_scaling || grads->d_rotation;
    const bool all = grads->d_xyz && grads->d_f_dc && grads->d_opacity && grads->d_scaling && grads->d_rotation;
    if (any && !all) return fail(-2, "Gaussian gradient outputs must be all set or all NULL");
    if (grads->max_radii2D && (!grads->grad_accum || !grads->denom)) return fail(-2, "statistics outputs must be all set or all NULL");
    sg.d_xyz = grads->d_xyz; sg.d_f_dc = grads->d_f_dc; sg.d_opacity = grads->d_opacity; sg.d_scaling = grads->d_scaling;
    sg.d_rotation = grads->d_rotation; sg.max_radii2D = grads->max_radii2D; sg.grad_accum = grads->grad_accum; sg.denom = grads->denom;
    sg.d_f_rest = (all && in->sh_degree > 0) ? grads->d_f_rest : nullptr;
    if (all && in->sh_degree > 0 && !grads->d_f_rest) return fail(-2, "sh_degree > 0: the gradient outputs need d_f_rest too");
  }
  PoseAdam pa = {};
  if (pose_adam && pose_adam->pose) {
    if (!pose_adam->m || !pose_adam->v || !pose_adam->step) return fail(-2, "pose Adam state missing");
    pa.pose = pose_adam->pose; pa.m = pose_adam->m; pa.v = pose_adam->v; pa.step = pose_adam->step;
    pa.lr_q = pose_adam->lr_q; pa.lr_t = pose_adam->lr_t; pa.beta1 = pose_adam->beta1; pa.beta2 = pose_adam->beta2; pa.eps = (float)pose_adam->eps;
    if (pose_adam->prior_pose && (pose_adam->prior_w_t != 0.f || pose_adam->prior_w_q != 0.f)) {
      pa.prior = pose_adam->prior_pose; pa.prior_w_t = pose_adam->prior_w_t; pa.prior_w_q = pose_adam->prior_w_q;
    }
    pa.best = pose_adam->best;
  }
  MapAdam ma;
  memset(&ma, 0, sizeof(ma));
  if (map_adam)
    if (int rc2 = map_adam_dev(map_adam, ma)) return rc2;
  const bool sh = in->sh_degree > 0 && in->f_rest;
  if (sh && ma.on && !ma.rp) return fail(-2, "sh_degree > 0: the map Adam state needs the f_rest group");
  // (an active SH degree > 0: the pose gradient runs through the colours' viewing direction, so even a tracking iteration needs the colour sums of
  //  the mapping-layout records)
  const bool tracking = sg.d_xyz == nullptr && !ma.on && !sh;
  if (tl && !tracking && !tl->dmaps) return fail(-1, "internal: a loss folded into the mapping backward needs the SSIM maps");
  if (pose_chain && !tracking) return fail(-1, "internal: the pose chain is a tracking-mode path");
  if (!compositor_done)
  { ProfScope ps(tracking ? MM3DGS_PROF_COMPOSITE_BWD_TRACK : MM3DGS_PROF_COMPOSITE_BWD, s);
    launch_composite_bwd_slam(cd, tracking, g, iv, b, N_capacity, dL_dout, bw.dsub, s, tl, dl_planes, pose_chain); }
  if (pose_chain) {
    // the tracking compositor applied the pose chain per (block, splat) and left one pose-gradient row per TILE at the head of the scratch: no
    // gradient records, no backward projection -- the pose finish sums the rows (GeomView.poserec, composite.hip)
    ProfScope ps(MM3DGS_PROF_PREPROCESS_BWD, s);
    launch_slam_pose_finish(bw.dsub, cd.gx * cd.gy, in->pose, dL_dpose, pa, s, pls.rows ? &pls : nullptr, prior_loss4, &iv.hdr->overflow);
    return check_launch("slam_backward");
  }
  // mapping run, direct bins, in-kernel Adam, no pose step: this launch also projects and bins the NEXT iteration's view
  const bool fuse = fuse_next_pose && !tracking && ma.on && db_bwd.on && !dL_dpose && !pa.pose && !sg.d_xyz && !sh;
  if (fused_out) *fused_out = fuse;
  if (fuse) {
    ProfScope ps(MM3DGS_PROF_PREPROCESS_BWD, s);
    launch_slam_bwd_project(cd, P, slam_in(in), (int32_t*)radii, g, image_view((void*)image_state, cd.H, cd.W), b, N_capacity, bw, sg, ma, fuse_next_pose,
                            db_bwd.bin_cap, db_bwd.rec_cap, db_bwd.slot_bits, s);
  } else
  { ProfScope ps(MM3DGS_PROF_PREPROCESS_BWD, s); launch_slam_preprocess_bwd(cd, P, slam_in(in), radii, g, b, N_capacity, bw, sg, dL_dpose, pa, ma, s, pls.rows ? &pls : nullptr, prior_loss4, db_bwd.on, &iv.hdr->overflow); }
  return check_launch("slam_backward");
}

int mm3dgs_slam_backward(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, const int32_t* radii,
                         const void* geom_state, const void* image_state, const void* binning_state, size_t N_capacity,
                         const float* dL_dout, void* backward_scratch, const Mm3dgsSlamGrads* grads, float* dL_dpose,
                         const Mm3dgsPoseAdam* pose_adam, const Mm3dgsMapAdam* map_adam, int flags, void* stream) {
  return slam_backward_impl(cam, P, in, radii, geom_state, image_state, binning_state, N_capacity, dL_dout, backward_scratch, grads, dL_dpose,
                            pose_adam, map_adam, flags, stream, nullptr);
}

static size_t loss_rows(int H, int W) { return (size_t)((W + 15) / 16) * ((H + 15) / 16); }
static LossCfg loss_cfg_dev(const Mm3dgsLossConfig* c) {
  LossCfg lc;
  lc.H = c->H; lc.W = c->W; lc.w_l1 = c->w_l1; lc.w_ssim = c->w_ssim; lc.w_pearson = c->w_pearson; lc.l1_mask = c->l1_mask;
  lc.pearson_mask = c->pearson_mask; lc.pearson_invert = c->pearson_invert; lc.sil_thr = c->sil_thr;
  for (int i = 0; i < 11; i++) lc.window[i] = c->window[i];
  lc.w_depth = c->w_depth_l1; lc.depth_mask = c->depth_l1_mask; lc.l1_sum = c->l1_sum;
  return lc;
}
// the splatam forms of the losses (depth-L1 term, colour L1 over { ref > 0 }, sums instead of means) exist in the standalone loss
// kernels only: a configuration that uses them is never folded into the compositors
static bool loss_is_variant(const Mm3dgsLossConfig* c) { return c->w_depth_l1 != 0.f || c->l1_sum != 0 || (c->l1_mask & 2) != 0; }
static int loss_cfg_check(const Mm3dgsLossConfig* c, const float* ref) {
  if (c->w_pearson != 0.f && !ref) return fail(-2, "Pearson term needs a reference depth");
  if ((c->w_depth_l1 != 0.f || (c->l1_mask & 2)) && !ref) return fail(-2, "depth-L1 term / { ref > 0 } mask needs a reference depth");
  if (c->w_depth_l1 != 0.f && c->w_pearson != 0.f) return fail(-2, "the depth-L1 and Pearson terms are exclusive");
  return 0;
}
size_t mm3dgs_loss_work_bytes(int H, int W) {
  return 256 + align_up((size_t)9 * H * W * 4, 256) + align_up(loss_rows(H, W) * 12 * 8, 256);
}

int mm3dgs_loss(const Mm3dgsLossConfig* c, const float* out6, const float* gt_color, const float* ref, void* work, float* dL,
                float* loss4, void* stream) {
  if (!c || !out6 || !gt_color || !work || !dL) return fail(-1, "NULL argument");
  if (c->H <= 0 || c->W <= 0) return fail(-1, "bad image size");
  if ((double)c->H * c->W * 36.0 >= 4294967296.0) return fail(-1, "image too large for the loss kernels' 32-bit offsets");
  if (int rc = loss_cfg_check(c, ref)) return rc;
  LossCfg lc = loss_cfg_dev(c);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MM3DGS_PROF_LOSS, s);
  char* w = (char*)work;
  launch_loss(lc, out6, gt_color, ref, (float*)(w + 256), (double*)w, (double*)(w + 256 + align_up((size_t)9 * c->H * c->W * 4, 256)), dL,
              loss4, s);
  return check_launch("loss");
}

int mm3dgs_slam_track(int n_iter, const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, float* out_color, int32_t* radii,
                      void* geom_state, void* image_state, void* binning_state, size_t N_capacity, int fwd_flags,
                      const Mm3dgsLossConfig* loss_cfg, const float* gt_color, const float* ref, void* loss_work, float* dL_dout,
                      float* loss4, void* backward_scratch, const Mm3dgsPoseAdam* pose_adam, void* stream) {
  if (n_iter < 0) return fail(-1, "n_iter < 0");
  if (!pose_adam || !pose_adam->pose) return fail(-2, "tracking needs the pose Adam state");
  Mm3dgsSlamGrads none;
  memset(&none, 0, sizeof(none));
  if (!loss_cfg || !gt_color || !loss_work || !dL_dout) return fail(-1, "NULL argument");
  if (int rc = loss_cfg_check(loss_cfg, ref)) return rc;
  // without SSIM every loss term is per pixel: fold the loss into the compositors (two launches and the gradient image
  // round trip less per iteration); needs the sort + forward-composite kernel (its workgroup = one 16x16 loss tile)
  const int no_fold = env_flag("MM3DGS_NO_FOLDED_LOSS", 0);   // read per call: tests compare both paths in one process
  const bool sh = in->sh_degree > 0 && in->f_rest;      // (active SH degree > 0: standalone loss kernels, mapping-mode compositor, no pose chain -- see slam_backward_impl)
  const bool fold = !no_fold && !sh && !loss_is_variant(loss_cfg) && loss_cfg->w_ssim == 0.f && slam_fused_sort(fwd_flags) && cam->image_height > 0 && cam->image_width > 0;
  TrackLoss tl = {};
  if (fold) {
    tl.cfg = loss_cfg_dev(loss_cfg);
    char* w = (char*)loss_work;
    tl.gt = gt_color; tl.ref = ref; tl.out = out_color; tl.sums = (const double*)w;
    tl.partial = (double*)(w + 256 + align_up((size_t)9 * loss_cfg->H * loss_cfg->W * 4, 256));
    tl.loss4 = loss4;
    tl.defer_scale = loss_cfg->w_pearson == 0.f ? 1 : 0;   // masked L1 only: no loss-finish launch, 1/n goes to the pose gradient
    if (loss_cfg->H != cam->image_height || loss_cfg->W != cam->image_width) return fail(-1, "loss and camera image sizes differ");
  }
  // masked L1 alone (normalisation deferred to the pose gradient): the backward compositor runs in the forward launch
  const bool fuse_track = fold && tl.defer_scale && !env_flag("MM3DGS_NO_FUSED_TRACK", 0) && backward_scratch;
  float* track_dsub = nullptr;
  if (fuse_track) {
    track_dsub = bwd_view(backward_scratch, P, N_capacity).dsub;
  }
  // round 6: the pose chain -- the projection writes every splat's { Kp, Kq, x }, the tracking compositor applies it per (block, splat) and leaves one
  // pose row per tile: no gradient records, no per-tile combine, no backward-projection launch (three launches per iteration).  The shipped mode only
  // (means pre-transformed: the world-frame mode's pose gradient also runs through the covariance rotation); MM3DGS_NO_POSE_CHAIN keeps the record path
  // (read per call: tests compare both in one process)
  const bool pose_chain = composite_has_pose_chain() && !in->world_means && !sh && backward_scratch && !env_flag("MM3DGS_NO_POSE_CHAIN", 0) &&
                          bwd_bytes_impl(P, N_capacity) >= (size_t)tiles_x(cam->image_width) * tiles_y(cam->image_height) * 32 * sizeof(float);
  if (n_iter > 0) slam_refresh_tile_order(cam, image_state, fwd_flags, stream);
  for (int it = 0; it < n_iter; it++) {
    int rc = slam_forward_impl(cam, P, in, out_color, radii, geom_state, image_state, binning_state, N_capacity, fwd_flags, stream, fold ? &tl : nullptr, track_dsub, false, pose_chain);
    if (rc) return rc;
    if (fold) {
      if (!tl.defer_scale) {
        ProfScope ps(MM3DGS_PROF_LOSS, (hipStream_t)stream);
        launch_loss_finish(tl.cfg, (double*)loss_work, tl.partial, (hipStream_t)stream);
      }
    } else {
      rc = mm3dgs_loss(loss_cfg, out_color, gt_color, ref, loss_work, dL_dout, loss4, stream);
      if (rc) return rc;
    }
    rc = slam_backward_impl(cam, P, in, radii, geom_state, image_state, binning_state, N_capacity, dL_dout, backward_scratch, &none,
                            nullptr, pose_adam, nullptr, fwd_flags, stream, fold ? &tl : nullptr, loss4, 6, fuse_track, nullptr, nullptr, pose_chain);
    if (rc) return rc;
  }
  return 0;
}

int mm3dgs_slam_map(int n_iter, const Mm3dgsMapView* views, const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, float* out_color,
                    int32_t* radii, void* geom_state, void* image_state, void* binning_state, size_t N_capacity, int fwd_flags,
                    const Mm3dgsLossConfig* loss_cfg, void* loss_work, float* dL_dout, float* loss4, void* backward_scratch,
                    const Mm3dgsSlamGrads* grads_stats, const Mm3dgsMapAdam* map_adam, void* stream) {
  if (n_iter < 0) return fail(-1, "n_iter < 0");
  if (n_iter > 0 && (!views || !in)) return fail(-1, "NULL argument");
  if (n_iter > 0 && !map_adam && !(grads_stats && grads_stats->d_xyz)) return fail(-1, "neither an Adam state nor gradient outputs");
  Mm3dgsSlamGrads sg;
  memset(&sg, 0, sizeof(sg));
  if (grads_stats) sg = *grads_stats;
  Mm3dgsMapAdam ad;
  memset(&ad, 0, sizeof(ad));
  if (map_adam) ad = *map_adam;
  Mm3dgsSlamInputs si = *in;
  if (n_iter > 0 && (!loss_cfg || !loss_work || !dL_dout)) return fail(-1, "NULL argument");
  // SSIM losses on the fused-sort path (the shipped mapping loss): the forward compositor's epilogue writes the per-tile L1 /
  // Pearson rows, the SSIM kernel's extra workgroup reduces them, and the gradient image has four planes: two launches per
  // iteration for the loss instead of three (no finishing launch), the loss values only once, at the end of the run
  const int no_rows = env_flag("MM3DGS_NO_FORWARD_ROWS", 0);   // read per call: tests compare both paths in one process
  const bool rows = n_iter > 0 && !no_rows && !loss_is_variant(loss_cfg) && loss_cfg->w_ssim != 0.f && slam_fused_sort(fwd_flags) && loss_cfg->H == cam->image_height &&
                    loss_cfg->W == cam->image_width;
  TrackLoss tl = {};
  LossCfg lc = {};
  char* w = (char*)loss_work;
  double* sums = (double*)w;
  float* dmaps = (float*)(w + 256);
  double* partial = n_iter > 0 ? (double*)(w + 256 + align_up((size_t)9 * loss_cfg->H * loss_cfg->W * 4, 256)) : nullptr;
  if (rows) {
    lc = loss_cfg_dev(loss_cfg);
    tl.cfg = lc; tl.out = out_color; tl.sums = sums; tl.partial = partial; tl.loss4 = nullptr; tl.defer_scale = 0;
  }
  // the backward launch of iteration `it` projects and bins the view of iteration `it + 1` when nothing stands between them (direct bins,
  // in-kernel Adam, neither view steps its pose); MM3DGS_NO_FUSED_PROJECT keeps the two launches apart (tests compare both, bit for bit)
  const int no_fuse_proj = env_flag("MM3DGS_NO_FUSED_PROJECT", 0);
  bool projected = (fwd_flags & MM3DGS_FWD_PROJECTED) != 0;     // (view 0 only: mm3dgs_slam_adam_project launched its projection + binning)
  if (n_iter > 0) slam_refresh_tile_order(cam, image_state, fwd_flags, stream);
  for (int it = 0; it < n_iter; it++) {
    if (!views[it].pose || !views[it].gt_color) return fail(-1, "view %d: NULL pose or colour target", it);
    if (loss_cfg->w_pearson != 0.f && !views[it].ref_depth_or_null) return fail(-2, "view %d: Pearson term needs a reference depth", it);
    if ((loss_cfg->w_depth_l1 != 0.f || (loss_cfg->l1_mask & 2)) && !views[it].ref_depth_or_null) return fail(-2, "view %d: depth-L1 term needs a reference depth", it);
    si.pose = views[it].pose;
    int rc;
    if (rows) {
      tl.gt = views[it].gt_color; tl.ref = views[it].ref_depth_or_null;
      rc = slam_forward_impl(cam, P, &si, out_color, radii, geom_state, image_state, binning_state, N_capacity, fwd_flags, stream, &tl, nullptr, projected);
      if (rc) return rc;
      // the gradient-image pass itself runs in the backward compositor's prologue (tl.dmaps set) unless MM3DGS_NO_FOLDED_LOSS asks
      // for the separate launch
      const bool fold_grad = !env_flag("MM3DGS_NO_FOLDED_LOSS", 0);
      { ProfScope ps(MM3DGS_PROF_LOSS, (hipStream_t)stream);
        launch_loss_after_forward_rows(lc, out_color, tl.gt, tl.ref, dmaps, sums, partial, fold_grad ? nullptr : dL_dout, (hipStream_t)stream); }
      if (loss4 && it == n_iter - 1) launch_loss_finish(lc, sums, partial, (hipStream_t)stream, loss4);
      rc = check_launch("loss");
      if (rc) return rc;
      tl.dmaps = fold_grad ? dmaps : nullptr;
      if (views[it].pose_adam_or_null && views[it].dpose_out_or_null) return fail(-2, "view %d: either a pose step or a pose-gradient output", it);
      const float* next_pose = (!no_fuse_proj && map_adam && it + 1 < n_iter && !views[it].pose_adam_or_null && !views[it + 1].pose_adam_or_null &&
                                !views[it].dpose_out_or_null && views[it + 1].pose) ? views[it + 1].pose : nullptr;
      projected = false;
      rc = slam_backward_impl(cam, P, &si, radii, geom_state, image_state, binning_state, N_capacity, dL_dout, backward_scratch, &sg, views[it].dpose_out_or_null,
                              views[it].pose_adam_or_null, map_adam ? &ad : nullptr, fwd_flags, stream, fold_grad ? &tl : nullptr, nullptr, 4, false,
                              next_pose, &projected);
    } else {
      // (MM3DGS_FWD_PROJECTED speaks of view 0 only)
      rc = mm3dgs_slam_forward(cam, P, &si, out_color, radii, geom_state, image_state, binning_state, N_capacity,
                               it == 0 ? fwd_flags : (fwd_flags & ~MM3DGS_FWD_PROJECTED), stream);
      if (rc) return rc;
      rc = mm3dgs_loss(loss_cfg, out_color, views[it].gt_color, views[it].ref_depth_or_null, loss_work, dL_dout, loss4, stream);
      if (rc) return rc;
      rc = mm3dgs_slam_backward(cam, P, &si, radii, geom_state, image_state, binning_state, N_capacity, dL_dout, backward_scratch, &sg,
                                views[it].dpose_out_or_null, views[it].pose_adam_or_null, map_adam ? &ad : nullptr, fwd_flags, stream);
    }
    if (rc) return rc;
    ad.step++;
  }
  return 0;
}

int mm3dgs_adam(const Mm3dgsAdamGroup* groups, int n_groups, int step, double beta1, double beta2, double eps, void* stream) {
  if (!groups || n_groups < 0 || n_groups > 8) return fail(-1, "bad group table");
  if (step < 1) return fail(-1, "step must be >= 1");
  AdamArgs a;
  a.ngroups = n_groups; a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  a.bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
  for (int i = 0; i < n_groups; i++) {
    if (groups[i].n && (!groups[i].param || !groups[i].grad || !groups[i].exp_avg || !groups[i].exp_avg_sq)) return fail(-1, "NULL in group %d", i);
    a.grp[i].p = groups[i].param; a.grp[i].g = groups[i].grad; a.grp[i].m = groups[i].exp_avg; a.grp[i].v = groups[i].exp_avg_sq;
    a.grp[i].n = groups[i].n; a.grp[i].step_size = (float)(groups[i].lr / bc1);
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MM3DGS_PROF_ADAM, s);
  launch_fused_adam(a, s);
  return check_launch("adam");
}

int mm3dgs_slam_direct_bins(const Mm3dgsCamera* cam, int P, size_t N_capacity, int fwd_flags) {
  if (!cam || P <= 0 || cam->image_height <= 0 || cam->image_width <= 0) return 0;
  return slam_direct_bins(fwd_flags, cam_dev(cam), P, N_capacity).on ? 1 : 0;
}

int mm3dgs_slam_adam_project(const Mm3dgsCamera* cam, int P, const Mm3dgsSlamInputs* in, const Mm3dgsSlamGrads* grads, const Mm3dgsMapAdam* adam,
                             int32_t* radii, void* geom_state, void* image_state, void* binning_state, size_t N_capacity, int fwd_flags, void* stream) {
  int rc = check_slam(cam, P, in);
  if (rc) return rc;
  if (in->sh_degree > 0) return fail(-2, "mm3dgs_slam_adam_project has no SH form: step with mm3dgs_adam, the next mm3dgs_slam_map call projects");
  if (!grads || !adam || !geom_state || !image_state || !binning_state || (P > 0 && !radii)) return fail(-1, "NULL argument");
  if (!grads->d_xyz || !grads->d_f_dc || !grads->d_opacity || !grads->d_scaling || !grads->d_rotation) return fail(-2, "all five gradient arrays are needed");
  hipStream_t s = (hipStream_t)stream;
  CamDev cd = cam_dev(cam);
  cd.bg_extras = 1;
  cd.tile_table = slam_tile_table(fwd_flags) ? 1 : 0;
  const DirectBins db = slam_direct_bins(fwd_flags, cd, P, N_capacity);
  if (!db.on) return fail(-3, "mm3dgs_slam_adam_project needs direct bins (flags / map size / capacity)");
  cd.trec_cap = db.trec_cap;
  MapAdam ma;
  memset(&ma, 0, sizeof(ma));
  if ((rc = map_adam_dev(adam, ma))) return rc;
  SlamGrads sg = {};
  sg.d_xyz = grads->d_xyz; sg.d_f_dc = grads->d_f_dc; sg.d_opacity = grads->d_opacity; sg.d_scaling = grads->d_scaling; sg.d_rotation = grads->d_rotation;
  { ProfScope ps(MM3DGS_PROF_ADAM, s);
    launch_slam_adam_project(cd, P, slam_in(in), radii, geom_view(geom_state, P > 0 ? P : 1), image_view(image_state, cd.H, cd.W), bin_view(binning_state, N_capacity),
                             sg, ma, in->pose, db.bin_cap, db.rec_cap, db.slot_bits, s); }
  return check_launch("slam_adam_project");
}

// ---- map surgery (compact.hip) -----------------------------------------------------------------------------------------
int mm3dgs_covisibility_ratio(int H, int W, const float* depth, const float* silhouette, const float* keyframe_pose, const float* current_pose,
                              float fx, float fy, float cx, float cy, uint32_t* counts, void* stream) {
  if (H < 0 || W < 0 || !depth || !silhouette || !keyframe_pose || !current_pose || !counts) return fail(-1, "covisibility_ratio: bad argument");
  launch_covisibility_ratio(H, W, depth, silhouette, keyframe_pose, current_pose, fx, fy, cx, cy, counts, (hipStream_t)stream);
  return check_launch("covisibility_ratio");
}

int mm3dgs_propagate_const_vel(const float* pose_m1, const float* pose_m2, float* out_pose, void* stream) {
  if (!pose_m1 || !pose_m2 || !out_pose) return fail(-1, "propagate_const_vel: NULL argument");
  launch_propagate_const_vel(pose_m1, pose_m2, out_pose, (hipStream_t)stream);
  return check_launch("propagate_const_vel");
}

int mm3dgs_prune_mask(int P, const float* opacity, const float* log_scales, const float* max_radii2D, float min_opacity, float max_scale,
                      float max_screen_size, uint8_t* keep, uint32_t* n_pruned_accum, void* stream) {
  if (P < 0) return fail(-1, "P < 0");
  if (P > 0 && (!opacity || !log_scales || !keep || !n_pruned_accum)) return fail(-1, "NULL argument");
  launch_prune_mask(P, opacity, log_scales, max_radii2D, min_opacity, max_scale, max_screen_size, max_radii2D ? 1 : 0, keep, n_pruned_accum,
                    (hipStream_t)stream);
  return check_launch("prune_mask");
}
size_t mm3dgs_compact_work_bytes(size_t n) { return align_up(((n + 255) / 256 + 1) * 4, 256); }
int mm3dgs_compact_plan(size_t n, const uint8_t* keep, void* work, uint32_t* n_keep, void* stream) {
  if (n > 0x7fffffffull) return fail(-1, "n too large");
  if (!work || !n_keep || (n > 0 && !keep)) return fail(-1, "NULL argument");
  launch_compact_plan((int)n, keep, (uint32_t*)work, n_keep, (hipStream_t)stream);
  return check_launch("compact_plan");
}
int mm3dgs_compact_rows(size_t n, const uint8_t* keep, const void* work, const Mm3dgsCompactArray* arrays, int n_arrays, void* stream) {
  if (n > 0x7fffffffull) return fail(-1, "n too large");
  if (n_arrays < 0 || n_arrays > 32) return fail(-1, "at most 32 arrays per call");
  if (n > 0 && (!keep || !work || (n_arrays > 0 && !arrays))) return fail(-1, "NULL argument");
  CompactTable t;
  t.n_arrays = 0;
  for (int a = 0; a < n_arrays; a++) {
    if (arrays[a].width < 0) return fail(-1, "array %d: negative width", a);
    if (arrays[a].width == 0) continue;     // empty rows (e.g. f_rest at SH degree 0)
    if (!arrays[a].src || !arrays[a].dst) return fail(-1, "array %d: NULL pointer", a);
    t.src[t.n_arrays] = arrays[a].src; t.dst[t.n_arrays] = arrays[a].dst; t.width[t.n_arrays] = arrays[a].width; t.n_arrays++;
  }
  launch_compact_rows((int)n, keep, (const uint32_t*)work, t, (hipStream_t)stream);
  return check_launch("compact_rows");
}
int mm3dgs_seed_gaussians(int H, int W, const float* color, const float* depth, const uint8_t* keep, const void* work, const float* pose, float fx,
                          float fy, float cx, float cy, uint32_t row0, const Mm3dgsSeedOutputs* out, int n_rest, void* stream) {
  if (H <= 0 || W <= 0 || (double)H * W > 2147483647.0) return fail(-1, "bad image size");
  if (!color || !depth || !keep || !work || !pose || !out) return fail(-1, "NULL argument");
  if (!out->xyz || !out->f_dc || !out->opacity || !out->scaling || !out->rotation || !out->rgb || (n_rest > 0 && !out->f_rest))
    return fail(-1, "NULL output array");
  SeedOut o = {out->xyz, out->f_dc, out->f_rest, out->opacity, out->scaling, out->rotation, out->rgb, n_rest > 0 ? n_rest : 0};
  launch_seed_gaussians(H, W, color, depth, keep, (const uint32_t*)work, pose, fx, fy, cx, cy, row0, o, (hipStream_t)stream);
  return check_launch("seed_gaussians");
}

int mm3dgs_mark_visible(const Mm3dgsCamera* cam, int P, const float* means3D, uint8_t* visible, void* stream) {
  if (!cam || !cam->viewmatrix || (P > 0 && (!means3D || !visible))) return fail(-1, "NULL buffer");
  launch_mark_visible(cam_dev(cam), P, means3D, visible, (hipStream_t)stream);
  return check_launch("mark_visible");
}

}  // extern "C"
