// depth.hip -- next row N4: partial-view scene clouds from a depth camera.
// The reference renders each problem with PyBullet from a fixed camera pose, removes the robot's pixels and
// back-projects the depth image (run_inference.py:194-257, robofin Bullet.get_pointcloud_from_camera), then draws
// 4096 of the points without replacement (run_inference.py:78-85).  Here the primitives are ray-cast
// analytically, one thread per pixel, for a whole batch: the nearest hit among cuboids, cylinders and (optionally)
// the robot's collision spheres; pixels whose nearest hit is the robot -- or nothing -- are dropped.  The
// subset is chosen on the device too: every valid pixel gets a Philox key, the n_out smallest keys win (a uniform
// subset in uniform order), found by a three-level radix select and sorted in LDS.
// PyBullet rasterises meshes and quantises depth; this is the exact-geometry equivalent (parity unpinned).
#include "depth_device.h"
#include "philox.h"
#include "select_device.h"

enum { STREAM_DEPTH = 9 };

// depth[b, v*W + u] = distance along the pixel's ray to the nearest obstacle surface, -1 if none / robot first
__global__ void __launch_bounds__(256)
    depth_render_kernel(const float *__restrict__ cam, float fx, float fy, float cx, float cy, int W, int H,
                        const float *__restrict__ cub_f, const float *__restrict__ cub_d, int M1,
                        const float *__restrict__ cyl_f, const float *__restrict__ cyl_r,
                        const float *__restrict__ cyl_h, int M2, const float *__restrict__ sph_c,
                        const float *__restrict__ sph_r, int S, float far_clip, float *__restrict__ depth) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= W * H) return;
  const float *P = cam + 16 * (size_t)b;
  const float ox = P[3], oy = P[7], oz = P[11];
  float dx, dy, dz;
  mpx_pixel_ray(P, fx, fy, cx, cy, pix % W, pix / W, dx, dy, dz);
  float best = far_clip;
  for (int m = 0; m < M1; ++m) {
    const size_t pm = (size_t)b * M1 + m;
    const float a0 = cub_d[3 * pm], a1 = cub_d[3 * pm + 1], a2 = cub_d[3 * pm + 2];
    if (mpx_is_zero(a0) || mpx_is_zero(a1) || mpx_is_zero(a2)) continue;
    best = fminf(best, ray_cuboid(cub_f + 16 * pm, a0 / 2.0f, a1 / 2.0f, a2 / 2.0f, ox, oy, oz, dx, dy, dz));
  }
  for (int m = 0; m < M2; ++m) {
    const size_t pm = (size_t)b * M2 + m;
    if (mpx_is_zero(cyl_r[pm]) || mpx_is_zero(cyl_h[pm])) continue;
    best = fminf(best, ray_cylinder(cyl_f + 16 * pm, cyl_r[pm], cyl_h[pm] / 2.0f, ox, oy, oz, dx, dy, dz));
  }
  bool robot = false;
  for (int s = 0; s < S; ++s) {
    const float *c = sph_c + ((size_t)b * S + s) * 3;
    if (ray_sphere(c[0], c[1], c[2], sph_r[s], ox, oy, oz, dx, dy, dz) < best) robot = true;
  }
  depth[(size_t)b * W * H + pix] = (robot || !(best < far_clip)) ? -1.0f : best;
}

// One workgroup per environment: picks n_out of the valid pixels (smallest Philox keys; ties by pixel id) and
// writes their world points in key order.  count[b] = number of valid pixels; if it is < n_out nothing is written.
__global__ void __launch_bounds__(SEL_THREADS)
    depth_select_kernel(const float *__restrict__ depth, const float *__restrict__ cam, float fx, float fy, float cx,
                        float cy, int W, int H, int n_out, uint32_t k0, uint32_t k1, uint32_t env0, float *__restrict__ out,
                        int64_t obs, int ops, int32_t *__restrict__ count) {
  __shared__ unsigned long long sel[SEL_CAP];
  __shared__ int hist[2048];
  __shared__ int s3[3];
  const int b = blockIdx.x, tid = threadIdx.x, HW = W * H;
  const float *dp = depth + (size_t)b * HW;
  const int valid = mpx_select_smallest(
      HW, n_out,
      [&](int g, uint32_t (&key)[4], bool (&valid)[4]) {  // one Philox block keys four consecutive pixels
        const Philox r = philox4x32((uint32_t)g, env0 + (uint32_t)b, STREAM_DEPTH, 0u, k0, k1);
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = r.c[u], valid[u] = 4 * g + u < HW && dp[min(4 * g + u, HW - 1)] >= 0.0f;
      },
      sel, hist, s3);
  if (tid == 0) count[b] = valid;
  if (valid < n_out) return;  // np.random.choice would raise: the host reports it
  const float *P = cam + 16 * (size_t)b;
  for (int i = tid; i < n_out; i += SEL_THREADS) {
    const int pix = (int)(uint32_t)sel[i];
    float dx, dy, dz;
    mpx_pixel_ray(P, fx, fy, cx, cy, pix % W, pix / W, dx, dy, dz);
    const float s = dp[pix];
    float *o = out + (int64_t)b * obs + (int64_t)i * ops;
    o[0] = mpx_fma(s, dx, P[3]);
    o[1] = mpx_fma(s, dy, P[7]);
    o[2] = mpx_fma(s, dz, P[11]);
  }
}

MPX_EXPORT int mpx_depth_render(const float *cam_poses, float fx, float fy, float cx, float cy, int W, int H, int B,
                                const float *cub_frames, const float *cub_dims, int M1, const float *cyl_frames,
                                const float *cyl_radii, const float *cyl_heights, int M2, const float *sph_centers,
                                const float *sph_radii, int S, float far_clip, float *depth, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && B <= 65535 && W > 0 && H > 0 && M1 >= 0 && M2 >= 0 && S >= 0, "mpx_depth_render: bad size");
  MPX_REQUIRE(fx > 0 && fy > 0 && far_clip > 0, "mpx_depth_render: bad intrinsics");
  if (B == 0) return 0;
  hipLaunchKernelGGL(depth_render_kernel, dim3(cdiv((int64_t)W * H, 256), B), dim3(256), 0, mpx_s(stream), cam_poses,
                     fx, fy, cx, cy, W, H, cub_frames, cub_dims, M1, cyl_frames, cyl_radii, cyl_heights, M2,
                     sph_centers, sph_radii, S, far_clip, depth);
  MPX_LAUNCH_CHECK("mpx_depth_render");
}

MPX_EXPORT int mpx_depth_select(const float *depth, const float *cam_poses, float fx, float fy, float cx, float cy,
                                int W, int H, int B, int n_out, uint64_t seed, int64_t env_offset, float *out,
                                int64_t out_batch_stride,
                                int out_point_stride, int32_t *count, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && W > 0 && H > 0, "mpx_depth_select: bad size");
  MPX_REQUIRE(n_out >= 1 && n_out <= SEL_MAX_OUT, "mpx_depth_select: n_out must be in [1, %d]", SEL_MAX_OUT);
  MPX_REQUIRE(out_point_stride >= 3 && count, "mpx_depth_select: bad output");
  MPX_REQUIRE(env_offset >= 0 && env_offset + B <= 0xFFFFFFFFll, "mpx_depth_select: env_offset + B exceeds 2^32");
  if (B == 0) return 0;
  hipLaunchKernelGGL(depth_select_kernel, dim3(B), dim3(SEL_THREADS), 0, mpx_s(stream), depth, cam_poses, fx, fy, cx,
                     cy, W, H, n_out, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)env_offset, out, out_batch_stride,
                     out_point_stride,
                     count);
  MPX_LAUNCH_CHECK("mpx_depth_select");
}
