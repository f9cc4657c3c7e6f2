// loss.hip -- next row N1: the training losses of mpinets/loss.py with
// their analytic gradients, forward + backward in one launch each.
//
//   collision_loss   (loss.py:48-95)   hinge(margin - min(cuboid sdf, cylinder sdf)) over a point cloud
//   point_match_loss (loss.py:31-45)   mse + l1 between two clouds
//   FrankaSampler.sample backward      d(points)/d(q) through the kinematic chain (robofin, autograd there)
//
// All three reduce per environment inside one workgroup in a fixed order (no atomics: results
// are run-to-run identical); the mean over the batch is left to the caller ([B] partial sums).
#include "sdf_device.h"



template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float *lds /* >= NV * waves */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float x = v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if (lane == 0) lds[wave * NV + i] = x;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float x = 0.0f;
    for (int w = 0; w < nw; ++w) x += lds[w * NV + threadIdx.x];
    lds[threadIdx.x] = x;  // only thread i reads slots w*NV+i, then writes slot i (i < NV): race free
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = lds[i];
  __syncthreads();
}

__device__ __forceinline__ float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// gradient of the 2-norm-of-positive-parts + clamped-max "box" distance w.r.t. d (n = 2 or 3);
// mirrors autograd of geometry.py:276-284 (norm has zero gradient at the origin; max picks the first index)
template <int ND>
__device__ __forceinline__ void box_grad(const float (&d)[ND], float (&g)[ND]) {
  float n2 = 0.0f;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    float m = fmaxf(d[i], 0.0f);
    n2 = mpx_fma(m, m, n2);
  }
  const float outside = sqrtf(n2);
  int arg = 0;
  float mx = d[0];
#pragma unroll
  for (int i = 1; i < ND; ++i)
    if (d[i] > mx) {
      mx = d[i];
      arg = i;
    }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    float gi = (outside > 0.0f && d[i] > 0.0f) ? d[i] / outside : 0.0f;
    if (mx < 0.0f && i == arg) gi += 1.0f;
    g[i] = gi;
  }
}

// ---- collision hinge ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    collision_hinge_kernel(const float *__restrict__ pts, int64_t pbs, int pps, int N,
                           const float *__restrict__ cub_f, const float *__restrict__ cub_d, int M1,
                           const float *__restrict__ cyl_f, const float *__restrict__ cyl_r,
                           const float *__restrict__ cyl_h, int M2, float margin, float *__restrict__ loss_sum,
                           float *__restrict__ grad, int64_t gbs, int gps) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float *cf = cub_f + (size_t)b * M1 * 16, *cd = cub_d + (size_t)b * M1 * 3;
  const float *yf = cyl_f + (size_t)b * M2 * 16, *yr = cyl_r + (size_t)b * M2, *yh = cyl_h + (size_t)b * M2;
  float acc[1] = {0.0f};
  for (int j = threadIdx.x; j < N; j += 256) {
    const float *p = pts + (int64_t)b * pbs + (int64_t)j * pps;
    const float x = p[0], y = p[1], z = p[2];
    float best = __builtin_inff();
    int arg = -1;  // < M1: cuboid, else cylinder; the reference's torch.min / torch.minimum keep the first minimum
    for (int m = 0; m < M1; ++m) {
      float s = cuboid_sdf(cf + 16 * m, cd[3 * m], cd[3 * m + 1], cd[3 * m + 2], x, y, z);
      if (s < best) best = s, arg = m;
    }
    for (int m = 0; m < M2; ++m) {
      float s = cylinder_sdf(yf + 16 * m, yr[m], yh[m], x, y, z);
      if (s < best) best = s, arg = M1 + m;
    }
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    const float h = margin - best;  // hinge_embedding_loss with target -1 (loss.py:89-94)
    if (h > 0.0f && arg >= 0) {
      acc[0] += h;
      if (grad) {
        const float *f = arg < M1 ? cf + 16 * arg : yf + 16 * (arg - M1);
        float px, py, pz, l0, l1, l2;
        mpx_project(f, x, y, z, px, py, pz);
        if (arg < M1) {
          float d[3] = {__builtin_fabsf(px) - cd[3 * arg] / 2.0f, __builtin_fabsf(py) - cd[3 * arg + 1] / 2.0f,
                        __builtin_fabsf(pz) - cd[3 * arg + 2] / 2.0f};
          float g[3];
          box_grad<3>(d, g);
          l0 = g[0] * sgn(px), l1 = g[1] * sgn(py), l2 = g[2] * sgn(pz);
        } else {
          const int m = arg - M1;
          const float rho = sqrtf(mpx_fma(py, py, px * px));
          float d[2] = {rho - yr[m], __builtin_fabsf(pz) - yh[m] / 2.0f};
          float g[2];
          box_grad<2>(d, g);
          const float ir = rho > 0.0f ? g[0] / rho : 0.0f;
          l0 = ir * px, l1 = ir * py, l2 = g[1] * sgn(pz);
        }
        // d(loss)/d(world point) = -(M^T g_local), M = rows 0..2 x cols 0..2 of the inverse frame
        gx = -(mpx_fma(f[8], l2, mpx_fma(f[4], l1, f[0] * l0)));
        gy = -(mpx_fma(f[9], l2, mpx_fma(f[5], l1, f[1] * l0)));
        gz = -(mpx_fma(f[10], l2, mpx_fma(f[6], l1, f[2] * l0)));
      }
    }
    if (grad) {
      float *g = grad + (int64_t)b * gbs + (int64_t)j * gps;
      g[0] = gx, g[1] = gy, g[2] = gz;
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) loss_sum[b] = acc[0];
}

// ---- point match ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    point_match_kernel(const float *__restrict__ a, const float *__restrict__ t, int n, float *__restrict__ sums,
                       float w_sq, float w_abs, float *__restrict__ grad) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  const float *pa = a + (size_t)b * n, *pt = t + (size_t)b * n;
  float acc[2] = {0.0f, 0.0f};
  for (int j = threadIdx.x; j < n; j += 256) {
    const float d = pa[j] - pt[j];
    acc[0] = mpx_fma(d, d, acc[0]);
    acc[1] += __builtin_fabsf(d);
    if (grad) grad[(size_t)b * n + j] = mpx_fma(w_sq * 2.0f, d, w_abs * sgn(d));
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) sums[2 * b] = acc[0], sums[2 * b + 1] = acc[1];
}

// ---- backward of the FK robot cloud ------------------------------------------------------------------------
// grad_q[b,k] = sum over points of  z_k . ((p - o_k) x g_p)   for the revolute joints k upstream of the
// point's link (joint k turns frame k+1 about that frame's z axis).
__global__ void __launch_bounds__(256)
    franka_cloud_grad_kernel(const float *__restrict__ q, float finger, const float *__restrict__ tpts,
                             const int32_t *__restrict__ tlink, const int32_t *__restrict__ subset, int n,
                             const float *__restrict__ gp, int64_t gbs, int gps, float *__restrict__ gq) {
  __shared__ float fr[15 * 12];
  __shared__ float red[7 * 4];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    float qq[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qq[j] = q[(size_t)b * 7 + j];
    franka_fk_frames(qq, finger, fr);
  }
  __syncthreads();
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = threadIdx.x; j < n; j += 256) {
    const int src = subset ? subset[j] : j;
    const int link = tlink[src];
    float px, py, pz;
    rigid_apply(fr + 12 * link, tpts[3 * (size_t)src], tpts[3 * (size_t)src + 1], tpts[3 * (size_t)src + 2], px, py, pz);
    const float *g = gp + (int64_t)b * gbs + (int64_t)j * gps;
    const float gx = g[0], gy = g[1], gz = g[2];
    const int nj = link < 7 ? link : 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (k < nj) {
        const float *f = fr + 12 * (k + 1);
        const float rx = px - f[9], ry = py - f[10], rz = pz - f[11];
        const float cx = ry * gz - rz * gy, cy = rz * gx - rx * gz, cz = rx * gy - ry * gx;
        acc[k] += mpx_fma(f[8], cz, mpx_fma(f[5], cy, f[2] * cx));
      }
    }
  }
  block_sum<7>(acc, red);
  if (threadIdx.x < 7) gq[(size_t)b * 7 + threadIdx.x] = acc[threadIdx.x];
}



MPX_EXPORT int mpx_collision_hinge(const float *points, int64_t batch_stride, int point_stride, int B, int N,
                                   const float *cub_frames, const float *cub_dims, int M1, const float *cyl_frames,
                                   const float *cyl_radii, const float *cyl_heights, int M2, float margin,
                                   float *loss_sum, float *grad_points, int64_t grad_batch_stride,
                                   int grad_point_stride, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N >= 0 && M1 >= 0 && M2 >= 0, "mpx_collision_hinge: negative size");
  MPX_REQUIRE(point_stride >= 3 && (!grad_points || grad_point_stride >= 3), "mpx_collision_hinge: point stride < 3");
  MPX_REQUIRE(loss_sum, "mpx_collision_hinge: loss_sum is NULL");
  if (B == 0) return 0;
  hipLaunchKernelGGL(collision_hinge_kernel, dim3(B), dim3(256), 0, mpx_s(stream), points, batch_stride, point_stride,
                     N, cub_frames, cub_dims, M1, cyl_frames, cyl_radii, cyl_heights, M2, margin, loss_sum,
                     grad_points, grad_batch_stride, grad_point_stride);
  MPX_LAUNCH_CHECK("mpx_collision_hinge");
}

MPX_EXPORT int mpx_point_match(const float *input, const float *target, int B, int n_per_env, float w_sq, float w_abs,
                               float *sums, float *grad_input, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && n_per_env >= 0, "mpx_point_match: negative size");
  MPX_REQUIRE(sums, "mpx_point_match: sums is NULL");
  if (B == 0) return 0;
  hipLaunchKernelGGL(point_match_kernel, dim3(B), dim3(256), 0, mpx_s(stream), input, target, n_per_env, sums, w_sq,
                     w_abs, grad_input);
  MPX_LAUNCH_CHECK("mpx_point_match");
}

MPX_EXPORT int mpx_franka_cloud_grad(const float *q, int B, float finger, const float *table_pts,
                                     const int32_t *table_link, const int32_t *subset, int n, const float *grad_points,
                                     int64_t grad_batch_stride, int grad_point_stride, float *grad_q,
                                     mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && n >= 0, "mpx_franka_cloud_grad: negative size");
  MPX_REQUIRE(grad_point_stride >= 3, "mpx_franka_cloud_grad: point stride < 3");
  if (B == 0) return 0;
  hipLaunchKernelGGL(franka_cloud_grad_kernel, dim3(B), dim3(256), 0, mpx_s(stream), q, finger, table_pts, table_link,
                     subset, n, grad_points, grad_batch_stride, grad_point_stride, grad_q);
  MPX_LAUNCH_CHECK("mpx_franka_cloud_grad");
}
