// Map surgery on the device (8f row 2): the pruning predicate, order-preserving stream compaction of the
// SoA state (parameters + Adam moments + densification statistics) and the seeding of new Gaussians from an RGB-D frame.
//
// Reference: slam/gaussian_model.py:380-451 (prune_points / _prune_optimizer: every per-Gaussian tensor keeps the rows with
// mask == false, in order), :574-588 (prune: sigmoid(opacity) < min_opacity, max exp(scaling) > 0.1 extent, max_radii2D > size
// threshold), slam/mapper.py:409-493,600-688 (one Gaussian per selected pixel in raster order: back-projected position,
// log-scale log(z / ((fx + fy) / 2)) on all axes, opacity logit 0, identity quaternion, f_dc = (rgb - 0.5) / C0).
//
// Compaction = three small launches over `n` flagged elements (Gaussians or pixels): per-256 counts, one workgroup scanning the
// counts, scatter by rank.  Order preserving and deterministic (no atomics decide a position).
#include "mm3dgs_common.h"
#include <algorithm>
#include "fused_api.h"

#define CB 256

// keep[i] = !pruned; *n_pruned += number of pruned elements (sticky device counter: the host may read it much later)
__global__ void __launch_bounds__(CB)
prune_mask_kernel(int P, const float* __restrict__ opacity, const float* __restrict__ scaling, const float* __restrict__ max_radii2D,
                  float min_opacity, float max_scale, float max_screen_size, int use_screen, uint8_t* __restrict__ keep,
                  uint32_t* __restrict__ n_pruned) {
  const int i = blockIdx.x * CB + threadIdx.x;
  bool pr = false;
  if (i < P) {
    const float o = 1.f / (1.f + expf(-opacity[i]));      // (accurate exp: the decisions must match torch.sigmoid / torch.exp)
    const float s = fmaxf(fmaxf(scaling[(size_t)i * 3], scaling[(size_t)i * 3 + 1]), scaling[(size_t)i * 3 + 2]);
    pr = (o < min_opacity) || (expf(s) > max_scale) || (use_screen && max_radii2D[i] > max_screen_size);
    keep[i] = pr ? 0 : 1;
  }
  const unsigned long long m = __ballot(pr);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_pruned, (uint32_t)__popcll(m));
}

__global__ void __launch_bounds__(CB) compact_count_kernel(int n, const uint8_t* __restrict__ keep, uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[CB / 64];
  const int i = blockIdx.x * CB + threadIdx.x;
  const unsigned long long m = __ballot(i < n && keep[i] != 0);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup: block_counts[0..nb) -> exclusive prefix in place, total to *n_keep
__global__ void __launch_bounds__(1024) compact_scan_kernel(int nb, uint32_t* __restrict__ block_counts, uint32_t* __restrict__ n_keep) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < nb ? block_counts[i] : 0u;
    const uint32_t x = wave_scan_incl(v);
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    uint32_t pre = carry;
    for (int w = 0; w < wv; w++) pre += wtot[w];
    if (i < nb) block_counts[i] = pre + x - v;
    __syncthreads();
    if (tid == 1023) carry = pre + x;
    __syncthreads();
  }
  if (tid == 0) *n_keep = carry;
}

// rank[i] = position of element i among the kept ones (only meaningful where keep[i])
__device__ __forceinline__ uint32_t compact_rank(int i, int n, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ block_pre,
                                                 bool& kept, uint32_t* wsum) {
  kept = i < n && keep[i] != 0;
  const unsigned long long m = __ballot(kept);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t pre = block_pre[blockIdx.x];
  for (int w = 0; w < wv; w++) pre += wsum[w];
  return pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

__global__ void __launch_bounds__(CB)
compact_rows_kernel(int n, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ block_pre, CompactTable t) {
  __shared__ uint32_t wsum[CB / 64];
  const int i = blockIdx.x * CB + threadIdx.x;
  bool kept;
  const uint32_t r = compact_rank(i, n, keep, block_pre, kept, wsum);
  if (!kept) return;
  for (int a = 0; a < t.n_arrays; a++) {
    const int w = t.width[a];
    const float* s = t.src[a] + (size_t)i * w;
    float* d = t.dst[a] + (size_t)r * w;
    for (int c = 0; c < w; c++) d[c] = s[c];
  }
}

// pixel i (raster order) with keep[i] seeds Gaussian row0 + rank(i)
__global__ void __launch_bounds__(CB)
seed_gaussians_kernel(int H, int W, const float* __restrict__ color, const float* __restrict__ depth, const uint8_t* __restrict__ keep,
                      const uint32_t* __restrict__ block_pre, const float* __restrict__ pose, float fx, float fy, float cx, float cy, uint32_t row0,
                      SeedOut o) {
  __shared__ uint32_t wsum[CB / 64];
  const int n = H * W;
  const int i = blockIdx.x * CB + threadIdx.x;
  bool kept;
  const uint32_t r = row0 + compact_rank(i, n, keep, block_pre, kept, wsum);
  if (!kept) return;
  // camera-to-world of the world-to-camera pose (qw,qx,qy,qz,tx,ty,tz): x_w = R^T (x_c - t)
  float qw = pose[0], qx = pose[1], qy = pose[2], qz = pose[3];
  const float qn = 1.f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
  qw *= qn; qx *= qn; qy *= qn; qz *= qn;
  const float R[3][3] = {{1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy)},
                         {2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx)},
                         {2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)}};
  const int u = i % W, v = i / W;
  const float z = depth[i];
  const float c[3] = {((float)u - cx) / fx * z - pose[4], ((float)v - cy) / fy * z - pose[5], z - pose[6]};
#pragma unroll
  for (int a = 0; a < 3; a++) o.xyz[(size_t)r * 3 + a] = R[0][a] * c[0] + R[1][a] * c[1] + R[2][a] * c[2];
  const float ls = logf(z / ((fx + fy) * 0.5f));     // log sqrt((z / f)^2)
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float rgb = color[(size_t)a * n + i];
    o.rgb[(size_t)r * 3 + a] = rgb;
    o.f_dc[(size_t)r * 3 + a] = (rgb - 0.5f) / 0.28209479177387814f;
    o.scaling[(size_t)r * 3 + a] = ls;
  }
  for (int a = 0; a < o.n_rest * 3; a++) o.f_rest[(size_t)r * o.n_rest * 3 + a] = 0.f;
  o.opacity[r] = 0.f;
  o.rotation[(size_t)r * 4] = 1.f; o.rotation[(size_t)r * 4 + 1] = 0.f; o.rotation[(size_t)r * 4 + 2] = 0.f; o.rotation[(size_t)r * 4 + 3] = 0.f;
}

// ---- keyframe test: covisibility ratio of two views (slam/mapper.py:141-173 need_new_keyframe -> get_depth_pointcloud :175-196 +
// is_covisible :198-216): the surface points of the last keyframe (its rendered depth where the silhouette is > 0.99, minus points
// that round to the world origin at 4 decimals) back-projected to the world and projected into the current view;
// counts[0] = points that land inside the image in front of the camera, counts[1] = points tested.  One pass over the rendered
// depth / silhouette planes instead of ~30 element-wise torch launches; a fixed small grid, so only a handful of atomics per counter.
__device__ __forceinline__ void pose_to_Rt(const float* __restrict__ pose, float R[3][3], float t[3]) {
  float w = pose[0], x = pose[1], y = pose[2], z = pose[3];
  const float n = 1.f / sqrtf(w * w + x * x + y * y + z * z);
  w *= n; x *= n; y *= n; z *= n;
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - w * z); R[0][2] = 2.f * (x * z + w * y);
  R[1][0] = 2.f * (x * y + w * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - w * x);
  R[2][0] = 2.f * (x * z - w * y); R[2][1] = 2.f * (y * z + w * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
  t[0] = pose[4]; t[1] = pose[5]; t[2] = pose[6];
}
__global__ void __launch_bounds__(CB)
covisibility_ratio_kernel(int H, int W, const float* __restrict__ depth, const float* __restrict__ sil, const float* __restrict__ kf_pose,
                          const float* __restrict__ cur_pose, float fx, float fy, float cx, float cy, uint32_t* __restrict__ counts) {
  float Rk[3][3], tk[3], Rc[3][3], tc[3];
  pose_to_Rt(kf_pose, Rk, tk);
  pose_to_Rt(cur_pose, Rc, tc);
  const int n = H * W;
  uint32_t n_in = 0, n_sel = 0;
  for (int i = blockIdx.x * CB + threadIdx.x; i < n; i += gridDim.x * CB) {
    const float z = sil[i] > 0.99f ? depth[i] : 0.f;
    const float u = (float)(i % W), v = (float)(i / W);
    const float c[3] = {(u - cx) / fx * z, (v - cy) / fy * z, z};
    // camera -> world of the keyframe (closed-form rigid inverse: R^T x - R^T t), then world -> current camera
    const float tt[3] = {-(Rk[0][0] * tk[0] + Rk[1][0] * tk[1] + Rk[2][0] * tk[2]), -(Rk[0][1] * tk[0] + Rk[1][1] * tk[1] + Rk[2][1] * tk[2]),
                         -(Rk[0][2] * tk[0] + Rk[1][2] * tk[1] + Rk[2][2] * tk[2])};
    float pw[3], p[3];
#pragma unroll
    for (int a = 0; a < 3; a++) pw[a] = c[0] * Rk[0][a] + c[1] * Rk[1][a] + c[2] * Rk[2][a] + tt[a];
    // torch.round(pts, decimals=4).abs().sum(1) > 0  <=>  some coordinate does not round (half to even) to zero
    const bool sel = z > 0.f && (fabsf(pw[0] * 10000.f) > 0.5f || fabsf(pw[1] * 10000.f) > 0.5f || fabsf(pw[2] * 10000.f) > 0.5f);
#pragma unroll
    for (int a = 0; a < 3; a++) p[a] = pw[0] * Rc[a][0] + pw[1] * Rc[a][1] + pw[2] * Rc[a][2] + tc[a];
    const float zc = p[2] + 1e-5f;
    const float uu = (fx * p[0] + cx * p[2]) / zc, vv = (fy * p[1] + cy * p[2]) / zc;
    const bool inside = sel && uu < (float)W && uu > 0.f && vv < (float)H && vv > 0.f && zc > 0.f;
    n_in += inside ? 1u : 0u; n_sel += sel ? 1u : 0u;
  }
  __shared__ uint32_t red[2][CB / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n_in += __shfl_down(n_in, off, 64); n_sel += __shfl_down(n_sel, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = n_in; red[1][threadIdx.x >> 6] = n_sel; }
  __syncthreads();
  if (threadIdx.x < 2) {
    uint32_t t = 0;
    for (int w = 0; w < CB / 64; w++) t += red[threadIdx.x][w];
    if (t) atomicAdd(&counts[threadIdx.x], t);
  }
}

// ---- launchers (called from api.hip) ------------------------------------------------------------------------------------------
void launch_prune_mask(int P, const float* opacity, const float* scaling, const float* max_radii2D, float min_opacity, float max_scale,
                       float max_screen_size, int use_screen, uint8_t* keep, uint32_t* n_pruned, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(prune_mask_kernel, dim3((P + CB - 1) / CB), dim3(CB), 0, s, P, opacity, scaling, max_radii2D, min_opacity, max_scale,
                     max_screen_size, use_screen, keep, n_pruned);
}
void launch_compact_plan(int n, const uint8_t* keep, uint32_t* block_pre, uint32_t* n_keep, hipStream_t s) {
  const int nb = (n + CB - 1) / CB;
  if (n > 0) hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(CB), 0, s, n, keep, block_pre);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, n > 0 ? nb : 0, block_pre, n_keep);
}
void launch_compact_rows(int n, const uint8_t* keep, const uint32_t* block_pre, const CompactTable& t, hipStream_t s) {
  if (n <= 0 || t.n_arrays <= 0) return;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((n + CB - 1) / CB), dim3(CB), 0, s, n, keep, block_pre, t);
}
void launch_seed_gaussians(int H, int W, const float* color, const float* depth, const uint8_t* keep, const uint32_t* block_pre,
                           const float* pose, float fx, float fy, float cx, float cy, uint32_t row0, const SeedOut& o, hipStream_t s) {
  const int n = H * W;
  if (n <= 0) return;
  hipLaunchKernelGGL(seed_gaussians_kernel, dim3((n + CB - 1) / CB), dim3(CB), 0, s, H, W, color, depth, keep, block_pre, pose, fx, fy, cx, cy,
                     row0, o);
}
__global__ void zero_words_kernel(uint32_t* __restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0u;
}
void launch_covisibility_ratio(int H, int W, const float* depth, const float* sil, const float* kf_pose, const float* cur_pose, float fx, float fy,
                               float cx, float cy, uint32_t* counts, hipStream_t s) {
  // (a two-word hipMemsetAsync goes through the runtime's fill path: ~75 us before the next kernel starts, measured; a one-wave launch: 5)
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, s, counts, 2);
  if (H * W <= 0) return;
  const int nb = std::min((H * W + CB - 1) / CB, 64);
  hipLaunchKernelGGL(covisibility_ratio_kernel, dim3(nb), dim3(CB), 0, s, H, W, depth, sil, kf_pose, cur_pose, fx, fy, cx, cy, counts);
}

// ---- constant-velocity pose prediction (utils/pose_utils.py:203-216 propagate_const_vel; :352-383 get_camera_from_tensor /
// get_tensor_from_camera): W' = (W1 W2^-1) W1 for the last two world->camera poses, back to (q, t).  One lane, double precision (the
// tracker used to read the two poses back and do this on the host: a device drain at the head of every frame).  Same algebra as
// pose_utils.propagate_const_vel_np: normalised quaternion -> R, closed-form rigid inverse, best-conditioned matrix -> quaternion branch.
__global__ void propagate_const_vel_kernel(const float* __restrict__ pm1, const float* __restrict__ pm2, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double R1[3][3], R2[3][3], t1[3], t2[3];
  const float* src[2] = {pm1, pm2};
  for (int v = 0; v < 2; v++) {
    double (*R)[3] = v == 0 ? R1 : R2;
    double* t = v == 0 ? t1 : t2;
    double w = src[v][0], x = src[v][1], y = src[v][2], z = src[v][3];
    const double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    R[0][0] = 1.0 - 2.0 * (y * y + z * z); R[0][1] = 2.0 * (x * y - w * z); R[0][2] = 2.0 * (x * z + w * y);
    R[1][0] = 2.0 * (x * y + w * z); R[1][1] = 1.0 - 2.0 * (x * x + z * z); R[1][2] = 2.0 * (y * z - w * x);
    R[2][0] = 2.0 * (x * z - w * y); R[2][1] = 2.0 * (y * z + w * x); R[2][2] = 1.0 - 2.0 * (x * x + y * y);
    t[0] = src[v][4]; t[1] = src[v][5]; t[2] = src[v][6];
  }
  // step = W1 W2^-1: rotation A = R1 R2^T, translation a = t1 - A t2;  W' = step W1: rotation A R1, translation A t1 + a
  double A[3][3], a[3], m[3][3], tn[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = R1[i][0] * R2[j][0] + R1[i][1] * R2[j][1] + R1[i][2] * R2[j][2];
  for (int i = 0; i < 3; i++) a[i] = t1[i] - (A[i][0] * t2[0] + A[i][1] * t2[1] + A[i][2] * t2[2]);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) m[i][j] = A[i][0] * R1[0][j] + A[i][1] * R1[1][j] + A[i][2] * R1[2][j];
    tn[i] = A[i][0] * t1[0] + A[i][1] * t1[1] + A[i][2] * t1[2] + a[i];
  }
  const double four_sq[4] = {1.0 + m[0][0] + m[1][1] + m[2][2], 1.0 + m[0][0] - m[1][1] - m[2][2], 1.0 - m[0][0] + m[1][1] - m[2][2],
                             1.0 - m[0][0] - m[1][1] + m[2][2]};
  const double cand[4][4] = {{four_sq[0], m[2][1] - m[1][2], m[0][2] - m[2][0], m[1][0] - m[0][1]},
                             {m[2][1] - m[1][2], four_sq[1], m[1][0] + m[0][1], m[0][2] + m[2][0]},
                             {m[0][2] - m[2][0], m[1][0] + m[0][1], four_sq[2], m[1][2] + m[2][1]},
                             {m[1][0] - m[0][1], m[2][0] + m[0][2], m[2][1] + m[1][2], four_sq[3]}};
  int best = 0;
  double mag = sqrt(fmax(four_sq[0], 0.0));
  for (int k = 1; k < 4; k++) {
    const double mk = sqrt(fmax(four_sq[k], 0.0));
    if (mk > mag) { mag = mk; best = k; }       // (first maximum, like argmax)
  }
  const double den = 2.0 * fmax(mag, 0.1);
  for (int k = 0; k < 4; k++) out[k] = (float)(cand[best][k] / den);
  for (int k = 0; k < 3; k++) out[4 + k] = (float)tn[k];
}
void launch_propagate_const_vel(const float* pm1, const float* pm2, float* out, hipStream_t s) {
  hipLaunchKernelGGL(propagate_const_vel_kernel, dim3(1), dim3(64), 0, s, pm1, pm2, out);
}
