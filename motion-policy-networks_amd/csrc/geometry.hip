// geometry.hip -- primitive inverse frames and point-vs-primitive signed distances.
//
// Reference semantics: /root/reference/mpinets/geometry.py
//   TorchCuboids._init_frames :177-223, .sdf :238-288, .sdf_sequence :290-347
//   TorchCylinders._init_frames :409-454, .sdf :456-507, .sdf_sequence :509-568
//   TorchSpheres.sdf :87-102, .sdf_sequence :104-123
//
// The reference materialises a [B,M,(T,)N,4] projected-point tensor and boolean-index copies
// of it; here one thread owns one point, walks the environment's primitives (frames arrive
// through the scalar cache: the environment index is block-uniform) and keeps the running
// minimum in a register.  Algorithmic traffic: 12 B read + 4 B written per point, plus
// M*(64+12) B of primitive data per environment.
#include "common.h"

#include <stdarg.h>
#include <string.h>

// ---- error string (one per host thread) ------------------------------------------------------
static thread_local char g_err[512] = "";
void mpx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
MPX_EXPORT const char *mpx_last_error(void) { return g_err; }
MPX_EXPORT int mpx_version(void) { return 340; }  // 340: mpx_pool_wgrad / _scratch / mpx_pool_dgrad, mpx_linear_segmax / _bf16x3, mpx_pack_rows_ld / _grad_ld; 330: mpx_sa3_front_bf16x3 (+ _pack, _pack_size, _w3_pairs), the three measurement hooks declared, mpx_sa_mlp_bf16x3_factored refuses nsample > 128; 320: mpx_linear_dact, mpx_segment_max_grad_act, mpx_linear_bf16x3_dact, mpx_linear_wgrad_bf16x3, unaligned frames accepted by mpx_franka_collision; 310: sa3_pack may be NULL, MPX_VARIANT_UNIT_QUEUE; 300: mpx_set_variant, wants_order(nsample); 200: env_offset arguments, mpx_rollout
MPX_EXPORT int mpx_device_info(char *name, int name_len, int *cu_count, int *lds_bytes) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    mpx_set_error("mpx_device_info: no HIP device");
    return 1;
  }
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  return 0;
}

// ---- frames ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prim_frames_kernel(const float *__restrict__ centers,
                                                          const float *__restrict__ quats, int n,
                                                          float *__restrict__ frames) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float q0 = quats[4 * i + 0], q1 = quats[4 * i + 1], q2 = quats[4 * i + 2], q3 = quats[4 * i + 3];
  float c0 = centers[3 * i + 0], c1 = centers[3 * i + 1], c2 = centers[3 * i + 2];
  float nrm = sqrtf(mpx_fma(q3, q3, mpx_fma(q2, q2, mpx_fma(q1, q1, q0 * q0))));
  float w = q0 / nrm;
  float x = -(q1 / nrm);
  float y = -(q2 / nrm);
  float z = -(q3 / nrm);
  float xx = 2.0f * (x * x), yy = 2.0f * (y * y), zz = 2.0f * (z * z);
  float wx = (2.0f * w) * x, wy = (2.0f * w) * y, wz = (2.0f * w) * z;
  float xy = (2.0f * x) * y, xz = (2.0f * x) * z, yz = (2.0f * y) * z;
  float R[9];
  R[0] = (1.0f - yy) - zz; R[1] = xy - wz;          R[2] = xz + wy;
  R[3] = xy + wz;          R[4] = (1.0f - xx) - zz; R[5] = yz - wx;
  // geometry.py:213 writes `yz - wx` here as well (a correct inverse has `yz + wx`); kept.
  R[6] = xz - wy;          R[7] = yz - wx;          R[8] = (1.0f - xx) - yy;
  float4 *o = reinterpret_cast<float4 *>(frames + 16 * (size_t)i);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = R[3 * r + 0] * (-c0);
    acc = mpx_fma(R[3 * r + 1], -c1, acc);
    acc = mpx_fma(R[3 * r + 2], -c2, acc);
    o[r] = make_float4(R[3 * r + 0], R[3 * r + 1], R[3 * r + 2], acc);
  }
  o[3] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
}

MPX_EXPORT int mpx_prim_frames(const float *centers, const float *quats, int n, float *inv_frames,
                               mpx_stream_t stream) {
  MPX_REQUIRE(n >= 0, "mpx_prim_frames: n < 0");
  if (n == 0) return 0;
  hipLaunchKernelGGL(prim_frames_kernel, dim3(cdiv(n, 256)), dim3(256), 0, mpx_s(stream), centers, quats,
                     n, inv_frames);
  MPX_LAUNCH_CHECK("mpx_prim_frames");
}

// ---- per-primitive distance functions (shared with franka.hip through sdf_device.h) -----------
#include "sdf_device.h"

enum { PRIM_CUBOID = 0, PRIM_CYLINDER = 1, PRIM_SPHERE = 2 };

// grid (ceil(P/256), B)
template <int KIND>
__global__ void __launch_bounds__(256)
    prim_sdf_kernel(const float *__restrict__ frames,  // cuboid/cyl: [B,M,16]; sphere: centers [B,M,3]
                    const float *__restrict__ pa,      // cuboid: dims [B,M,3]; cyl/sphere: radii [B,M]
                    const float *__restrict__ pb,      // cyl: heights [B,M]
                    int M, const float *__restrict__ points, int P, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P) return;
  const float *p = points + ((size_t)b * P + n) * 3;
  const float x = p[0], y = p[1], z = p[2];
  float best = __builtin_inff();
  for (int m = 0; m < M; ++m) {
    const size_t pm = (size_t)b * M + m;
    float s;
    if (KIND == PRIM_CUBOID) {
      s = cuboid_sdf(frames + 16 * pm, pa[3 * pm + 0], pa[3 * pm + 1], pa[3 * pm + 2], x, y, z);
    } else if (KIND == PRIM_CYLINDER) {
      s = cylinder_sdf(frames + 16 * pm, pa[pm], pb[pm], x, y, z);
    } else {
      s = sphere_sdf(frames[3 * pm + 0], frames[3 * pm + 1], frames[3 * pm + 2], pa[pm], x, y, z);
    }
    best = s < best ? s : best;
  }
  out[(size_t)b * P + n] = best;
}

template <int KIND>
static int launch_sdf(const char *name, const float *frames, const float *pa, const float *pb, int B, int M,
                      const float *points, int P, float *out, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && M >= 0 && P >= 0, "%s: negative size", name);
  MPX_REQUIRE(B <= 65535 * 64, "%s: B too large", name);
  if (B == 0 || P == 0) return 0;
  // gridDim.y is limited to 65535: walk the batch in slabs
  for (int b0 = 0; b0 < B; b0 += 65535) {
    int nb = B - b0 < 65535 ? B - b0 : 65535;
    size_t po = (size_t)b0 * M;
    const float *f = frames ? frames + (KIND == PRIM_SPHERE ? 3 : 16) * po : nullptr;
    const float *a = pa ? pa + (KIND == PRIM_CUBOID ? 3 : 1) * po : nullptr;
    const float *bb = pb ? pb + po : nullptr;
    hipLaunchKernelGGL(prim_sdf_kernel<KIND>, dim3(cdiv(P, 256), nb), dim3(256), 0, mpx_s(stream), f, a, bb,
                       M, points + (size_t)b0 * P * 3, P, out + (size_t)b0 * P);
  }
  MPX_LAUNCH_CHECK(name);
}

MPX_EXPORT int mpx_cuboid_sdf(const float *inv_frames, const float *dims, int B, int M, const float *points,
                              int P, float *out, mpx_stream_t stream) {
  return launch_sdf<PRIM_CUBOID>("mpx_cuboid_sdf", inv_frames, dims, nullptr, B, M, points, P, out, stream);
}
MPX_EXPORT int mpx_cylinder_sdf(const float *inv_frames, const float *radii, const float *heights, int B,
                                int M, const float *points, int P, float *out, mpx_stream_t stream) {
  return launch_sdf<PRIM_CYLINDER>("mpx_cylinder_sdf", inv_frames, radii, heights, B, M, points, P, out,
                                   stream);
}
MPX_EXPORT int mpx_sphere_sdf(const float *centers, const float *radii, int B, int M, const float *points,
                              int P, float *out, mpx_stream_t stream) {
  return launch_sdf<PRIM_SPHERE>("mpx_sphere_sdf", centers, radii, nullptr, B, M, points, P, out, stream);
}
