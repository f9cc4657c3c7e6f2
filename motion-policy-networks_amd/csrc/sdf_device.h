// sdf_device.h -- single-primitive signed distances, evaluation order pinned to
// oracle/mpn_oracle.c (which is pinned to the reference's mpinets/geometry.py goldens).
#pragma once
#include "common.h"

// torch.isclose(x, 0) with the defaults (rtol 1e-5, atol 1e-8): |x| <= 1e-8
// (geometry.py:56, :155-157, :385-388)
__device__ __forceinline__ bool mpx_is_zero(float x) { return __builtin_fabsf(x) <= 1e-8f; }

// f: 4x4 row-major inverse frame (rows 0..2 used: [R | Rt])
__device__ __forceinline__ void mpx_project(const float *__restrict__ f, float x, float y, float z, float &px,
                                            float &py, float &pz) {
  float a0 = f[0] * x;
  a0 = mpx_fma(f[1], y, a0);
  a0 = mpx_fma(f[2], z, a0);
  float a1 = f[4] * x;
  a1 = mpx_fma(f[5], y, a1);
  a1 = mpx_fma(f[6], z, a1);
  float a2 = f[8] * x;
  a2 = mpx_fma(f[9], y, a2);
  a2 = mpx_fma(f[10], z, a2);
  px = a0 + f[3];
  py = a1 + f[7];
  pz = a2 + f[11];
}

// geometry.py:276-287 for an UNMASKED cuboid (the caller has tested the mask)
__device__ __forceinline__ float cuboid_sdf_live(const float *__restrict__ f, float dx, float dy, float dz,
                                                 float x, float y, float z);
// geometry.py:276-287; masked (zero-volume) cuboid -> +inf
__device__ __forceinline__ float cuboid_sdf(const float *__restrict__ f, float dx, float dy, float dz,
                                            float x, float y, float z) {
  if (mpx_is_zero(dx) || mpx_is_zero(dy) || mpx_is_zero(dz)) return __builtin_inff();
  return cuboid_sdf_live(f, dx, dy, dz, x, y, z);
}
__device__ __forceinline__ float cuboid_sdf_live(const float *__restrict__ f, float dx, float dy, float dz,
                                                 float x, float y, float z) {
  float px, py, pz;
  mpx_project(f, x, y, z, px, py, pz);
  float d0 = __builtin_fabsf(px) - dx / 2.0f;
  float d1 = __builtin_fabsf(py) - dy / 2.0f;
  float d2 = __builtin_fabsf(pz) - dz / 2.0f;
  float m0 = fmaxf(d0, 0.0f), m1 = fmaxf(d1, 0.0f), m2 = fmaxf(d2, 0.0f);
  float outside = sqrtf(mpx_fma(m2, m2, mpx_fma(m1, m1, m0 * m0)));
  float inside = fminf(fmaxf(d0, fmaxf(d1, d2)), 0.0f);
  return outside + inside;
}

// geometry.py:486-506 for an UNMASKED cylinder
__device__ __forceinline__ float cylinder_sdf_live(const float *__restrict__ f, float radius, float height,
                                                   float x, float y, float z);
// geometry.py:486-506; masked (zero radius or height) -> +inf
__device__ __forceinline__ float cylinder_sdf(const float *__restrict__ f, float radius, float height,
                                              float x, float y, float z) {
  if (mpx_is_zero(radius) || mpx_is_zero(height)) return __builtin_inff();
  return cylinder_sdf_live(f, radius, height, x, y, z);
}
__device__ __forceinline__ float cylinder_sdf_live(const float *__restrict__ f, float radius, float height,
                                                   float x, float y, float z) {
  float px, py, pz;
  mpx_project(f, x, y, z, px, py, pz);
  float rho = sqrtf(mpx_fma(py, py, px * px));
  float d0 = __builtin_fabsf(rho) - radius;
  float d1 = __builtin_fabsf(pz) - height / 2.0f;
  float m0 = fmaxf(d0, 0.0f), m1 = fmaxf(d1, 0.0f);
  float outside = sqrtf(mpx_fma(m1, m1, m0 * m0));
  float inside = fminf(fmaxf(d0, d1), 0.0f);
  return outside + inside;
}

// geometry.py:98-101
__device__ __forceinline__ float sphere_sdf(float cx, float cy, float cz, float radius, float x, float y,
                                            float z) {
  if (mpx_is_zero(radius)) return __builtin_inff();
  float dx = x - cx, dy = y - cy, dz = z - cz;
  return sqrtf(mpx_fma(dz, dz, mpx_fma(dy, dy, dx * dx))) - radius;
}
