// depth_device.h -- ray / primitive intersections for the partial-view (depth camera) scene cloud, shared in
// spirit with oracle/mpn_oracle.c (same operation order; -ffp-contract=off on both sides).
#pragma once
#include "sdf_device.h"

// camera ray of pixel (u, v): OpenGL camera frame (x right, y up, looks along -z), like the reference's evaluation
// poses (run_inference.py:215-243 give world-from-camera poses whose -z axis points at the scene).
// P = world-from-camera 4x4 row-major.  Returns the unit direction in world coordinates.
__host__ __device__ __forceinline__ void mpx_pixel_ray(const float *P, float fx, float fy, float cx, float cy, int u,
                                                      int v, float &dx, float &dy, float &dz) {
  const float xc = ((float)u + 0.5f - cx) / fx, yc = -(((float)v + 0.5f - cy) / fy), zc = -1.0f;
  float wx = P[0] * xc, wy = P[4] * xc, wz = P[8] * xc;
  wx = mpx_fma(P[1], yc, wx), wy = mpx_fma(P[5], yc, wy), wz = mpx_fma(P[9], yc, wz);
  wx = mpx_fma(P[2], zc, wx), wy = mpx_fma(P[6], zc, wy), wz = mpx_fma(P[10], zc, wz);
  const float n = sqrtf(mpx_fma(wz, wz, mpx_fma(wy, wy, wx * wx)));
  dx = wx / n, dy = wy / n, dz = wz / n;
}

// rotate a direction / transform a point into a primitive's frame (rows 0..2 of the inverse frame)
__host__ __device__ __forceinline__ void mpx_rotate(const float *f, float x, float y, float z, float &ox, float &oy,
                                                   float &oz) {
  ox = mpx_fma(f[2], z, mpx_fma(f[1], y, f[0] * x));
  oy = mpx_fma(f[6], z, mpx_fma(f[5], y, f[4] * x));
  oz = mpx_fma(f[10], z, mpx_fma(f[9], y, f[8] * x));
}

constexpr float MPX_RAY_EPS = 1e-12f;

// nearest entry distance along the ray (camera outside the box), +inf if missed
__host__ __device__ __forceinline__ float ray_cuboid(const float *f, float hx, float hy, float hz, float ox, float oy,
                                                    float oz, float dx, float dy, float dz) {
  float lo[3], ld[3];
  mpx_project(f, ox, oy, oz, lo[0], lo[1], lo[2]);
  mpx_rotate(f, dx, dy, dz, ld[0], ld[1], ld[2]);
  const float h[3] = {hx, hy, hz};
  float tn = -__builtin_inff(), tf = __builtin_inff();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (__builtin_fabsf(ld[a]) < MPX_RAY_EPS) {
      if (__builtin_fabsf(lo[a]) > h[a]) return __builtin_inff();
    } else {
      const float t1 = (-h[a] - lo[a]) / ld[a], t2 = (h[a] - lo[a]) / ld[a];
      tn = fmaxf(tn, fminf(t1, t2));
      tf = fminf(tf, fmaxf(t1, t2));
    }
  }
  return (tn <= tf && tn > 0.0f) ? tn : __builtin_inff();
}

__host__ __device__ __forceinline__ float ray_cylinder(const float *f, float r, float hh, float ox, float oy, float oz,
                                                      float dx, float dy, float dz) {
  float lx, ly, lz, ex, ey, ez;
  mpx_project(f, ox, oy, oz, lx, ly, lz);
  mpx_rotate(f, dx, dy, dz, ex, ey, ez);
  float best = __builtin_inff();
  const float a = mpx_fma(ey, ey, ex * ex);
  if (a > MPX_RAY_EPS) {
    const float b = mpx_fma(ly, ey, lx * ex), c = mpx_fma(ly, ly, lx * lx) - r * r;
    const float disc = b * b - a * c;
    if (disc >= 0.0f) {
      const float s = (-b - sqrtf(disc)) / a;
      if (s > 0.0f && __builtin_fabsf(mpx_fma(s, ez, lz)) <= hh) best = s;
    }
  }
  if (__builtin_fabsf(ez) > MPX_RAY_EPS) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float s = ((k ? -hh : hh) - lz) / ez;
      const float px = mpx_fma(s, ex, lx), py = mpx_fma(s, ey, ly);
      if (s > 0.0f && s < best && mpx_fma(py, py, px * px) <= r * r) best = s;
    }
  }
  return best;
}

__host__ __device__ __forceinline__ float ray_sphere(float cx, float cy, float cz, float r, float ox, float oy, float oz,
                                                    float dx, float dy, float dz) {
  const float mx = ox - cx, my = oy - cy, mz = oz - cz;
  const float b = mpx_fma(mz, dz, mpx_fma(my, dy, mx * dx));
  const float c = mpx_fma(mz, mz, mpx_fma(my, my, mx * mx)) - r * r;
  const float disc = b * b - c;
  if (disc < 0.0f) return __builtin_inff();
  const float s = -b - sqrtf(disc);
  return s > 0.0f ? s : __builtin_inff();
}
