// scene.hip -- batched scene point clouds from cuboid / cylinder primitives, on the device.
//
// Replaces construct_mixed_point_cloud (/root/reference/mpinets/geometry.py:571-608) for batches of
// environments (closed-loop re-rendering, BASELINE config 5; data_loader.py:237-260 per sample).
// The reference: areas -> proportions; pool of n_i = int(p_i*N) + 500 surface samples per
// obstacle (geometry.py:598-604); np.random.choice(sum n_i, N, replace=False) (:608); labels are a
// shuffled 1..K (:594-595).  Its samples come from NumPy's global RNG inside geometrout, so only
// the DISTRIBUTION can be matched.  This kernel draws from exactly that distribution with a counter
// RNG (Philox4x32-10 keyed by (seed, GLOBAL environment id = env_offset + row): a shard of a larger batch draws
// exactly what the unsharded batch draws for the same environments):
//   * picking N of the pool slots without replacement in random order: every pool slot gets a
//     Philox key, the N smallest keys win and their order is the output order (one workgroup per
//     environment: radix select + LDS sort, select_device.h); only the obstacle owning each slot
//     matters, written as obstacle ids [B,N] (uint16);
//   * pool samples are i.i.d., so each output point is a fresh uniform sample on its obstacle's
//     surface: one thread per point, written straight into the slab rows (x,y,z[,label]).
// Zero-volume primitives are skipped exactly like data_loader.py:248,256 (is_zero_volume).
// Obstacle order = cuboids then cylinders (data_loader.py:258 `cuboids + cylinders`).
// The same algorithm is restated in oracle/mpn_oracle.c (orc_scene_*), bit-exact on the ids.
#include "common.h"
#include "philox.h"
#include "select_device.h"

constexpr int MAX_OBS = 96;
enum { STREAM_URN = 1, STREAM_LABEL = 2, STREAM_POINT = 3 };

// ---- kernel 1: per-environment allocation, label shuffle and the draw of N pool slots ----------------------
// one workgroup per environment
__device__ __forceinline__ double obstacle_area(int m, int M1, const float *cd, const float *yr, const float *yh) {
  if (m < M1) {
    const float *d = cd + 3 * m;
    if (__builtin_fabsf(d[0]) <= 1e-8f || __builtin_fabsf(d[1]) <= 1e-8f || __builtin_fabsf(d[2]) <= 1e-8f) return 0.0;
    return 2.0 * ((double)d[0] * d[1] + (double)d[0] * d[2] + (double)d[1] * d[2]);
  }
  const float r = yr[m - M1], h = yh[m - M1];
  if (__builtin_fabsf(r) <= 1e-8f || __builtin_fabsf(h) <= 1e-8f) return 0.0;
  return 2.0 * 3.14159265358979323846 * (double)r * (double)h + 2.0 * 3.14159265358979323846 * (double)r * (double)r;
}

__global__ void __launch_bounds__(SEL_THREADS)
    scene_assign_kernel(const float *__restrict__ cub_dims, int M1, const float *__restrict__ cyl_radii,
                        const float *__restrict__ cyl_heights, int M2, int B, int N, uint32_t seed_lo,
                        uint32_t seed_hi, uint32_t env0, uint16_t *__restrict__ assign, uint8_t *__restrict__ labels,
                        int32_t *__restrict__ n_obstacles) {
  __shared__ unsigned long long sel[SEL_CAP];
  __shared__ int hist[2048];
  __shared__ int s3[3];
  __shared__ int off_s[MAX_OBS + 1];  // pool offsets: obstacle m owns slots [off[m], off[m+1])
  __shared__ int K_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int M = M1 + M2;
  const float *cd = cub_dims + (size_t)b * M1 * 3, *yr = cyl_radii + (size_t)b * M2, *yh = cyl_heights + (size_t)b * M2;
  uint16_t *arow = assign + (size_t)b * N;
  __shared__ double area_s[MAX_OBS];
  __shared__ double total_s;
  __shared__ int pool_s[MAX_OBS];
  __shared__ uint8_t lab_s[MAX_OBS], who_s[MAX_OBS];  // label per obstacle; the live obstacles in order
  if (tid < M) area_s[tid] = obstacle_area(tid, M1, cd, yr, yh);  // double, like numpy; zero-volume -> 0
  __syncthreads();
  if (tid == 0) {
    double total = 0.0;
    for (int m = 0; m < M; ++m) total += area_s[m];
    total_s = total;
  }
  __syncthreads();
  if (tid < M) {
    const double a = area_s[tid];
    pool_s[tid] = a > 0.0 ? (int)((a / total_s) * (double)N) + 500 : 0;  // int(prop*num_points) + 500
  }
  __syncthreads();
  if (tid == 0) {
    int pool = 0, live = 0;
    for (int m = 0; m < M; ++m) {
      off_s[m] = pool;
      pool += pool_s[m];
      lab_s[m] = 0;
      if (pool_s[m]) {
        who_s[live] = (uint8_t)m;
        lab_s[m] = (uint8_t)(++live);
      }
    }
    off_s[M] = pool;
    K_s = live;
    if (n_obstacles) n_obstacles[b] = live;
    // labels: Fisher-Yates shuffle of 1..K over the live obstacles, walked from the back (random.shuffle,
    // geometry.py:594-595)
    uint32_t ctr = 0;
    for (int i = live - 1; i > 0; --i) {
      const Philox r = philox4x32(ctr++, env0 + (uint32_t)b, STREAM_LABEL, 0, seed_lo, seed_hi);
      const int jpos = (int)(((uint64_t)r.c[0] * (uint32_t)(i + 1)) >> 32);  // uniform in [0, i]
      const uint8_t tmp = lab_s[who_s[i]];
      lab_s[who_s[i]] = lab_s[who_s[jpos]];
      lab_s[who_s[jpos]] = tmp;
    }
  }
  __syncthreads();
  if (labels && tid < M) labels[(size_t)b * M + tid] = K_s ? lab_s[tid] : 0;
  __syncthreads();
  if (K_s == 0) {  // geometry.py:586-587 returns an empty cloud; ids 0xFFFF mark "no obstacle"
    for (int j = tid; j < N; j += SEL_THREADS) arow[j] = 0xFFFFu;
    return;
  }
  const int T = off_s[M];
  mpx_select_smallest(
      T, N,
      [&](int g, uint32_t (&key)[4], bool (&valid)[4]) {  // one Philox block keys four consecutive slots
        const Philox r = philox4x32((uint32_t)g, env0 + (uint32_t)b, STREAM_URN, 0, seed_lo, seed_hi);
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = r.c[u], valid[u] = 4 * g + u < T;
      },
      sel, hist, s3);  // T >= N + 500 always
  for (int j = tid; j < N; j += SEL_THREADS) {
    const int slot = (int)(uint32_t)sel[j];
    int m = 0;
    while (slot >= off_s[m + 1]) ++m;
    arow[j] = (uint16_t)m;
  }
}

// ---- kernel 2: one uniform surface sample per output point ----------------------------------------------
__device__ __forceinline__ void quat_rotate(const float *__restrict__ q, float x, float y, float z, float &ox,
                                            float &oy, float &oz) {
  // proper rotation by the normalised quaternion (w,x,y,z) -- poses of obstacles are true rigid
  // poses (geometrout); the non-orthonormal matrix of geometry.py:209-216 only exists in the SDF classes.
  const float n = sqrtf(mpx_fma(q[3], q[3], mpx_fma(q[2], q[2], mpx_fma(q[1], q[1], q[0] * q[0]))));
  const float w = q[0] / n, a = q[1] / n, b = q[2] / n, c = q[3] / n;
  const float r00 = 1.0f - 2.0f * (b * b + c * c), r01 = 2.0f * (a * b - w * c), r02 = 2.0f * (a * c + w * b);
  const float r10 = 2.0f * (a * b + w * c), r11 = 1.0f - 2.0f * (a * a + c * c), r12 = 2.0f * (b * c - w * a);
  const float r20 = 2.0f * (a * c - w * b), r21 = 2.0f * (b * c + w * a), r22 = 1.0f - 2.0f * (a * a + b * b);
  ox = mpx_fma(r02, z, mpx_fma(r01, y, r00 * x));
  oy = mpx_fma(r12, z, mpx_fma(r11, y, r10 * x));
  oz = mpx_fma(r22, z, mpx_fma(r21, y, r20 * x));
}

__global__ void __launch_bounds__(256)
    scene_points_kernel(const float *__restrict__ cub_c, const float *__restrict__ cub_d,
                        const float *__restrict__ cub_q, int M1, const float *__restrict__ cyl_c,
                        const float *__restrict__ cyl_r, const float *__restrict__ cyl_h,
                        const float *__restrict__ cyl_q, int M2, int N, uint32_t seed_lo, uint32_t seed_hi,
                        uint32_t env0, const uint16_t *__restrict__ assign, const uint8_t *__restrict__ labels,
                        float *__restrict__ out, int64_t obs, int ops, int write_label) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const int m = assign[(size_t)b * N + j];
  float *o = out + (int64_t)b * obs + (int64_t)j * ops;
  if (m == 0xFFFF) {
    o[0] = o[1] = o[2] = 0.0f;
    if (write_label) o[3] = 0.0f;
    return;
  }
  const Philox r = philox4x32((uint32_t)j, env0 + (uint32_t)b, STREAM_POINT, 0, seed_lo, seed_hi);
  const float u0 = u01(r.c[0]), u1 = u01(r.c[1]), u2 = u01(r.c[2]), u3 = u01(r.c[3]);
  float lx, ly, lz;
  const float *ctr, *quat;
  if (m < M1) {
    const float *d = cub_d + ((size_t)b * M1 + m) * 3;
    const float ax = d[1] * d[2], ay = d[0] * d[2], az = d[0] * d[1];
    const float t = u0 * (ax + ay + az);
    const float sgn = u1 < 0.5f ? -0.5f : 0.5f;
    // face pair by area, side by sign, the two free coordinates uniform over the face
    if (t < ax) {
      lx = sgn * d[0];
      ly = (u2 - 0.5f) * d[1];
      lz = (u3 - 0.5f) * d[2];
    } else if (t < ax + ay) {
      lx = (u2 - 0.5f) * d[0];
      ly = sgn * d[1];
      lz = (u3 - 0.5f) * d[2];
    } else {
      lx = (u2 - 0.5f) * d[0];
      ly = (u3 - 0.5f) * d[1];
      lz = sgn * d[2];
    }
    ctr = cub_c + ((size_t)b * M1 + m) * 3;
    quat = cub_q + ((size_t)b * M1 + m) * 4;
  } else {
    const int c = m - M1;
    const float rad = cyl_r[(size_t)b * M2 + c], h = cyl_h[(size_t)b * M2 + c];
    const float side = 2.0f * rad * h, cap = rad * rad;  // areas / pi
    const float t = u0 * (side + 2.0f * cap);
    float s, co;
    mpx_sincos(u1 * 6.28318530717958647692f, s, co);
    float rho = rad;
    lz = (u2 - 0.5f) * h;
    if (t >= side) {
      rho = rad * sqrtf(u3);
      lz = t < side + cap ? -0.5f * h : 0.5f * h;
    }
    lx = rho * co;
    ly = rho * s;
    ctr = cyl_c + ((size_t)b * M2 + c) * 3;
    quat = cyl_q + ((size_t)b * M2 + c) * 4;
  }
  float wx, wy, wz;
  quat_rotate(quat, lx, ly, lz, wx, wy, wz);
  o[0] = wx + ctr[0];
  o[1] = wy + ctr[1];
  o[2] = wz + ctr[2];
  if (write_label) o[3] = labels ? (float)labels[(size_t)b * (M1 + M2) + m] : 1.0f;
}

MPX_EXPORT int mpx_scene_cloud(const float *cub_centers, const float *cub_dims, const float *cub_quats, int M1,
                               const float *cyl_centers, const float *cyl_radii, const float *cyl_heights,
                               const float *cyl_quats, int M2, int B, int num_points, uint64_t seed,
                               int64_t env_offset, uint16_t *assign, uint8_t *labels, int32_t *n_obstacles, float *out,
                               int64_t out_batch_stride, int out_point_stride, int write_label,
                               mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && M1 >= 0 && M2 >= 0 && num_points >= 0, "mpx_scene_cloud: negative size");
  MPX_REQUIRE(M1 + M2 <= MAX_OBS, "mpx_scene_cloud: more than %d primitives per environment", MAX_OBS);
  MPX_REQUIRE(num_points <= SEL_MAX_OUT, "mpx_scene_cloud: num_points > %d (the draw is ordered in LDS)", SEL_MAX_OUT);
  MPX_REQUIRE(env_offset >= 0 && env_offset + B <= 0xFFFFFFFFll, "mpx_scene_cloud: env_offset + B exceeds 2^32");
  MPX_REQUIRE(out_point_stride >= (write_label ? 4 : 3), "mpx_scene_cloud: out_point_stride too small");
  MPX_REQUIRE(assign != nullptr, "mpx_scene_cloud: assign scratch [B,num_points] uint16 is required");
  if (B == 0 || num_points == 0) return 0;
  if (B > MPX_GRID_Y) {  // more environments than one launch's gridDim.y: slabs (the draws are keyed by env_offset + row)
    for (int64_t b0 = 0; b0 < B; b0 += MPX_GRID_Y) {
      const int nb = (int)(B - b0 < MPX_GRID_Y ? B - b0 : MPX_GRID_Y);
      if (int rc = mpx_scene_cloud(cub_centers + b0 * M1 * 3, cub_dims + b0 * M1 * 3, cub_quats + b0 * M1 * 4, M1,
                                   cyl_centers + b0 * M2 * 3, cyl_radii + b0 * M2, cyl_heights + b0 * M2, cyl_quats + b0 * M2 * 4,
                                   M2, nb, num_points, seed, env_offset + b0, assign + b0 * num_points,
                                   labels ? labels + b0 * (M1 + M2) : nullptr, n_obstacles ? n_obstacles + b0 : nullptr,
                                   out + b0 * out_batch_stride, out_batch_stride, out_point_stride, write_label, stream))
        return rc;
    }
    return 0;
  }
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32), env0 = (uint32_t)env_offset;
  hipLaunchKernelGGL(scene_assign_kernel, dim3(B), dim3(SEL_THREADS), 0, mpx_s(stream), cub_dims, M1, cyl_radii,
                     cyl_heights, M2, B, num_points, lo, hi, env0, assign, labels, n_obstacles);
  hipLaunchKernelGGL(scene_points_kernel, dim3(cdiv(num_points, 256), B), dim3(256), 0, mpx_s(stream), cub_centers,
                     cub_dims, cub_quats, M1, cyl_centers, cyl_radii, cyl_heights, cyl_quats, M2, num_points, lo,
                     hi, env0, assign, labels, out, out_batch_stride, out_point_stride, write_label);
  MPX_LAUNCH_CHECK("mpx_scene_cloud");
}
