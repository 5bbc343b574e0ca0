// Device-side maths shared by the per-Gaussian kernels (preprocess.hip, fused.hip): SH basis, quaternion -> rotation,
// 3D covariance, EWA projection to a 2D conic.  Conventions: see preprocess.hip's header.
#pragma once
#include "mm3dgs_common.h"

#define PP_BLOCK 256

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

// Real SH basis values b[0..(deg+1)^2) at unit direction (x,y,z).
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b) {
  b[0] = SH_C0;
  if (deg > 0) {
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
      b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
      if (deg > 2) {
        b[9] = SH_C3_0 * y * (3.f * xx - yy);
        b[10] = SH_C3_1 * xy * z;
        b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
        b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
        b[14] = SH_C3_5 * z * (xx - yy);
        b[15] = SH_C3_6 * x * (xx - 3.f * yy);
      }
    }
  }
}
// d b[k] / d(x,y,z)
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* gx, float* gy, float* gz) {
  gx[0] = gy[0] = gz[0] = 0.f;
  if (deg > 0) {
    gx[1] = 0.f; gy[1] = -SH_C1; gz[1] = 0.f;
    gx[2] = 0.f; gy[2] = 0.f; gz[2] = SH_C1;
    gx[3] = -SH_C1; gy[3] = 0.f; gz[3] = 0.f;
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      gx[4] = SH_C2_0 * y; gy[4] = SH_C2_0 * x; gz[4] = 0.f;
      gx[5] = 0.f; gy[5] = SH_C2_1 * z; gz[5] = SH_C2_1 * y;
      gx[6] = SH_C2_2 * -2.f * x; gy[6] = SH_C2_2 * -2.f * y; gz[6] = SH_C2_2 * 4.f * z;
      gx[7] = SH_C2_3 * z; gy[7] = 0.f; gz[7] = SH_C2_3 * x;
      gx[8] = SH_C2_4 * 2.f * x; gy[8] = SH_C2_4 * -2.f * y; gz[8] = 0.f;
      if (deg > 2) {
        gx[9] = SH_C3_0 * 6.f * xy; gy[9] = SH_C3_0 * (3.f * xx - 3.f * yy); gz[9] = 0.f;
        gx[10] = SH_C3_1 * yz; gy[10] = SH_C3_1 * xz; gz[10] = SH_C3_1 * xy;
        gx[11] = SH_C3_2 * -2.f * xy; gy[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); gz[11] = SH_C3_2 * 8.f * yz;
        gx[12] = SH_C3_3 * -6.f * xz; gy[12] = SH_C3_3 * -6.f * yz; gz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
        gx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); gy[13] = SH_C3_4 * -2.f * xy; gz[13] = SH_C3_4 * 8.f * xz;
        gx[14] = SH_C3_5 * 2.f * xz; gy[14] = SH_C3_5 * -2.f * yz; gz[14] = SH_C3_5 * (xx - yy);
        gx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); gy[15] = SH_C3_6 * -6.f * xy; gz[15] = 0.f;
      }
    }
  }
}

__device__ __forceinline__ void quat_to_R(const float* q, float R[3][3]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma3 (symmetric, full 3x3) from scale/rotation or from the 6 upper-triangular values.
__device__ __forceinline__ void load_cov3d(int idx, const float* scales, const float* rots, const float* cov3d,
                                           float mod, float S3[3][3], float R[3][3], float sm[3]) {
  if (cov3d) {
    const float* c = cov3d + (size_t)idx * 6;
    S3[0][0] = c[0]; S3[0][1] = c[1]; S3[0][2] = c[2];
    S3[1][0] = c[1]; S3[1][1] = c[3]; S3[1][2] = c[4];
    S3[2][0] = c[2]; S3[2][1] = c[4]; S3[2][2] = c[5];
  } else {
    float q[4] = {rots[(size_t)idx * 4], rots[(size_t)idx * 4 + 1], rots[(size_t)idx * 4 + 2], rots[(size_t)idx * 4 + 3]};
    quat_to_R(q, R);
    sm[0] = mod * scales[(size_t)idx * 3]; sm[1] = mod * scales[(size_t)idx * 3 + 1]; sm[2] = mod * scales[(size_t)idx * 3 + 2];
    float Mx[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) Mx[i][k] = R[i][k] * sm[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) S3[i][j] = Mx[i][0] * Mx[j][0] + Mx[i][1] * Mx[j][1] + Mx[i][2] * Mx[j][2];
  }
}

struct Ewa {
  float t[3];      // view-space mean
  float txc, tyc;  // clamped-frustum x,y used inside the Jacobian
  bool in_x, in_y;
  float A[2][3];   // J * Wr
  float J00, J02, J11, J12;
  float a, b, c;   // 2D covariance (+0.3 on the diagonal)
};

__device__ __forceinline__ void ewa_project(const CamDev& cam, const float* V, const float p[3], const float S3[3][3], Ewa& e) {
#pragma unroll
  for (int j = 0; j < 3; j++) e.t[j] = p[0] * V[0 * 4 + j] + p[1] * V[1 * 4 + j] + p[2] * V[2 * 4 + j] + V[3 * 4 + j];
  float tz = e.t[2];
  float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
  float txtz = e.t[0] / tz, tytz = e.t[1] / tz;
  e.in_x = (txtz >= -limx) && (txtz <= limx);
  e.in_y = (tytz >= -limy) && (tytz <= limy);
  e.txc = fminf(limx, fmaxf(-limx, txtz)) * tz;
  e.tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
  if (e.in_x) e.txc = e.t[0];
  if (e.in_y) e.tyc = e.t[1];
  float itz = 1.f / tz, itz2 = itz * itz;
  e.J00 = cam.focal_x * itz; e.J02 = -cam.focal_x * e.txc * itz2;
  e.J11 = cam.focal_y * itz; e.J12 = -cam.focal_y * e.tyc * itz2;
  // Wr[j][i] = V[i][j]  (world->view rotation for column vectors);  A = J * Wr
#pragma unroll
  for (int i = 0; i < 3; i++) {
    e.A[0][i] = e.J00 * V[i * 4 + 0] + e.J02 * V[i * 4 + 2];
    e.A[1][i] = e.J11 * V[i * 4 + 1] + e.J12 * V[i * 4 + 2];
  }
  float AS[2][3];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int j = 0; j < 3; j++) AS[r][j] = e.A[r][0] * S3[0][j] + e.A[r][1] * S3[1][j] + e.A[r][2] * S3[2][j];
  e.a = AS[0][0] * e.A[0][0] + AS[0][1] * e.A[0][1] + AS[0][2] * e.A[0][2] + 0.3f;
  e.b = AS[0][0] * e.A[1][0] + AS[0][1] * e.A[1][1] + AS[0][2] * e.A[1][2];
  e.c = AS[1][0] * e.A[1][0] + AS[1][1] * e.A[1][1] + AS[1][2] * e.A[1][2] + 0.3f;
}

