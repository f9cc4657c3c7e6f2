// batch.hip -- next row N2: training / validation batch assembly on the device.
// The reference builds every sample on a CPU DataLoader worker (mpinets/data_loader.py:141-280 `get_inputs`,
// :390-417 `__getitem__`): read the expert trajectory row from HDF5, add N(0, random_scale) joint noise and
// clamp to the limits (train only), normalise, FK the LAST waypoint for the target pose, then sample the three
// clouds.  Here the dataset arrays (HDF5 schema) live in HBM and one launch prepares the per-sample joint
// quantities for a whole batch; the clouds come from the kernels of the inference path
// (mpx_franka_cloud, mpx_pose_cloud, mpx_scene_cloud) writing straight into the [B,6272,4] slab.
#include "common.h"
#include "philox.h"

enum { STREAM_JOINT_NOISE = 7 };

// one thread per sample
__global__ void __launch_bounds__(64)
    batch_configs_kernel(const float *__restrict__ traj, int64_t n_traj, int L, const int64_t *__restrict__ traj_idx,
                         const int32_t *__restrict__ timestep, const float *__restrict__ limits, float noise_scale,
                         uint32_t seed_lo, uint32_t seed_hi, uint32_t sample0, int train, int B, float finger,
                         float *__restrict__ q,
                         float *__restrict__ q_norm, float *__restrict__ sup_norm, float *__restrict__ target_pose,
                         float *__restrict__ target_pos) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  int64_t ti = traj_idx[b];
  ti = ti < 0 ? 0 : (ti >= n_traj ? n_traj - 1 : ti);  // (an out-of-range index would read outside the array)
  int t = timestep ? timestep[b] : 0;
  t = t < 0 ? 0 : (t >= L ? L - 1 : t);                 // data_loader.py:403-404
  const int ts = t + 1 >= L ? L - 1 : t + 1;            // supervision: next waypoint, last one re-used (:408-412)
  const float *row = traj + (ti * L + t) * 7, *srow = traj + (ti * L + ts) * 7, *frow = traj + (ti * L + (L - 1)) * 7;
  float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (noise_scale > 0.0f) {  // Box-Muller on two Philox blocks keyed by (seed, sample)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const Philox r = philox4x32((uint32_t)blk, sample0 + (uint32_t)b, STREAM_JOINT_NOISE, 0u, seed_lo, seed_hi);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const float u1 = 1.0f - u01(r.c[2 * pr]), u2 = u01(r.c[2 * pr + 1]);  // u1 in (0,1]
        const float rad = sqrtf(-2.0f * logf(u1));
        float s, c;
        mpx_sincos(6.28318530717958647692f * u2, s, c);
        z[4 * blk + 2 * pr] = rad * c;
        z[4 * blk + 2 * pr + 1] = rad * s;
      }
    }
  }
  float fin[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const float lo = limits[2 * j], hi = limits[2 * j + 1];
    float v = row[j];
    if (noise_scale > 0.0f) v = noise_scale * z[j] + v;  // data_loader.py:169-171
    if (train) v = fminf(fmaxf(v, lo), hi);              // :176-178: every TRAIN sample is clamped, noise or not
    q[(size_t)b * 7 + j] = v;
    q_norm[(size_t)b * 7 + j] = (v - lo) / (hi - lo) * 2.0f + -1.0f;             // utils.py:91-93
    if (sup_norm) sup_norm[(size_t)b * 7 + j] = (srow[j] - lo) / (hi - lo) * 2.0f + -1.0f;
    fin[j] = frow[j];
  }
  float fr[15 * 12];
  franka_fk_frames(fin, finger, fr);                     // FrankaRealRobot.fk(last waypoint) (:155-157)
  const float *g = fr + 12 * 14;                         // right_gripper
  float *P = target_pose + (size_t)b * 16;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    P[4 * r + 0] = g[3 * r + 0];
    P[4 * r + 1] = g[3 * r + 1];
    P[4 * r + 2] = g[3 * r + 2];
    P[4 * r + 3] = g[9 + r];
    target_pos[(size_t)b * 3 + r] = g[9 + r];
  }
  P[12] = 0.0f, P[13] = 0.0f, P[14] = 0.0f, P[15] = 1.0f;
}

// rows of a [n, row] array gathered by index (primitives of the sampled trajectories)
__global__ void __launch_bounds__(256)
    gather_rows_kernel(const float *__restrict__ src, const int64_t *__restrict__ idx, int row, int64_t total,
                       float *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / row;
  dst[i] = src[idx[b] * row + (i - b * row)];
}

MPX_EXPORT int mpx_batch_configs(const float *trajectories, int64_t n_traj, int L, const int64_t *traj_idx,
                                 const int32_t *timestep, const float *limits, float noise_scale, uint64_t seed,
                                 int64_t sample_offset, int train, int B, float finger, float *q, float *q_norm, float *sup_norm, float *target_pose,
                                 float *target_pos, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && L >= 1 && n_traj >= 1, "mpx_batch_configs: bad size");
  MPX_REQUIRE(sample_offset >= 0 && sample_offset + B <= 0xFFFFFFFFll, "mpx_batch_configs: sample_offset + B exceeds 2^32");
  MPX_REQUIRE(trajectories && traj_idx && limits && q && q_norm && target_pose && target_pos,
              "mpx_batch_configs: NULL operand");
  if (B == 0) return 0;
  hipLaunchKernelGGL(batch_configs_kernel, dim3(cdiv(B, 64)), dim3(64), 0, mpx_s(stream), trajectories, n_traj, L,
                     traj_idx, timestep, limits, noise_scale, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sample_offset,
                     train, B, finger, q,
                     q_norm, sup_norm, target_pose, target_pos);
  MPX_LAUNCH_CHECK("mpx_batch_configs");
}

MPX_EXPORT int mpx_gather_rows(const float *src, const int64_t *idx, int B, int row_floats, float *dst,
                               mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && row_floats >= 0, "mpx_gather_rows: bad size");
  const int64_t total = (int64_t)B * row_floats;
  if (total == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, mpx_s(stream), src, idx, row_floats,
                     total, dst);
  MPX_LAUNCH_CHECK("mpx_gather_rows");
}
