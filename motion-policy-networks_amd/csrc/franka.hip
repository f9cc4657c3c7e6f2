// franka.hip -- Franka Panda forward kinematics, robot point cloud, collision spheres and the
// fused swept-sphere collision check.
//
// Replaces (reference call sites; the implementations live in un-vendored robofin v0.0.1):
//   FrankaSampler.sample              mpinets/model.py:250, run_inference.py:64,111,169
//   FrankaSampler.sample_end_effector run_inference.py:66-69, data_loader.py:158-161
//   FrankaSampler.end_effector_pose   mpinets/model.py:275
//   FrankaCollisionSampler.compute_spheres + the SDF sweep   mpinets/model.py:293-314
//   joint update of the rollout       mpinets/model.py:171-173, utils.py:207-209
//
// FK is a 7-step dependent chain (~650 VALU instructions) -- far more than the ~12
// instructions a table point needs -- so a workgroup computes FK for EPB configurations on EPB
// lanes at once (one instruction stream), parks the 15 frames of each in LDS and then all 256
// threads stream table points through them.
#include "common.h"
#include "sdf_device.h"
#include "philox.h"
#include "select_device.h"

constexpr int FRAME_FLOATS = MPX_NUM_FRAMES * 12;  // 180

// ---- FK frames to global -----------------------------------------------------------------------
__global__ void __launch_bounds__(64) franka_fk_kernel(const float *__restrict__ q, int B, float finger,
                                                       float *__restrict__ frames) {
  __shared__ float lds[64 * FRAME_FLOATS];
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b < B) {
    float qq[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qq[j] = q[(size_t)b * 7 + j];
    franka_fk_frames(qq, finger, lds + threadIdx.x * FRAME_FLOATS);
  }
  __syncthreads();
  // coalesced copy-out
  const int nb = min(64, B - blockIdx.x * 64);
  float *dst = frames + (size_t)blockIdx.x * 64 * FRAME_FLOATS;
  for (int i = threadIdx.x; i < nb * FRAME_FLOATS; i += 64) dst[i] = lds[i];
}

MPX_EXPORT int mpx_franka_fk(const float *q, int B, float finger, float *frames, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0, "mpx_franka_fk: B < 0");
  if (B == 0) return 0;
  hipLaunchKernelGGL(franka_fk_kernel, dim3(cdiv(B, 64)), dim3(64), 0, mpx_s(stream), q, B, finger, frames);
  MPX_LAUNCH_CHECK("mpx_franka_fk");
}

// ---- table points moved by FK frames -------------------------------------------------------------
constexpr int CLOUD_EPB = 16;  // configurations per workgroup

__global__ void __launch_bounds__(256)
    franka_cloud_kernel(const float *__restrict__ q, int B, float finger, const float *__restrict__ tpts,
                        const int32_t *__restrict__ tlink, const int32_t *__restrict__ subset, int n_out,
                        float *__restrict__ out, int64_t obs, int ops) {
  __shared__ float lds[CLOUD_EPB * FRAME_FLOATS];
  const int b0 = blockIdx.x * CLOUD_EPB;
  const int nb = min(CLOUD_EPB, B - b0);
  if (threadIdx.x < nb) {
    float qq[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qq[j] = q[(size_t)(b0 + threadIdx.x) * 7 + j];
    franka_fk_frames(qq, finger, lds + threadIdx.x * FRAME_FLOATS);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n_out; j += 256) {
    const int src = subset ? subset[j] : j;
    const float px = tpts[3 * (size_t)src + 0], py = tpts[3 * (size_t)src + 1], pz = tpts[3 * (size_t)src + 2];
    const int link = tlink[src];
    for (int e = 0; e < nb; ++e) {
      float ox, oy, oz;
      rigid_apply(lds + e * FRAME_FLOATS + 12 * link, px, py, pz, ox, oy, oz);
      float *o = out + (int64_t)(b0 + e) * obs + (int64_t)j * ops;
      o[0] = ox;
      o[1] = oy;
      o[2] = oz;
    }
  }
}

MPX_EXPORT int mpx_franka_cloud(const float *q, int B, float finger, const float *table_pts,
                                const int32_t *table_link, const int32_t *subset, int n_out, float *out,
                                int64_t out_batch_stride, int out_point_stride, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && n_out >= 0, "mpx_franka_cloud: negative size");
  MPX_REQUIRE(out_point_stride >= 3, "mpx_franka_cloud: out_point_stride < 3");
  if (B == 0 || n_out == 0) return 0;
  hipLaunchKernelGGL(franka_cloud_kernel, dim3(cdiv(B, CLOUD_EPB)), dim3(256), 0, mpx_s(stream), q, B,
                     finger, table_pts, table_link, subset, n_out, out, out_batch_stride, out_point_stride);
  MPX_LAUNCH_CHECK("mpx_franka_cloud");
}

MPX_EXPORT int mpx_franka_spheres(const float *q, int B, float finger, const float *sph_centers,
                                  const int32_t *sph_link, int S, float *out, mpx_stream_t stream) {
  return mpx_franka_cloud(q, B, finger, sph_centers, sph_link, nullptr, S, out, (int64_t)S * 3, 3, stream);
}

// ---- table points moved by one given pose per batch element ----------------------------------------
__global__ void __launch_bounds__(256)
    pose_cloud_kernel(const float *__restrict__ poses, const float *__restrict__ tpts,
                      const int32_t *__restrict__ subset, int n_out, float *__restrict__ out, int64_t obs,
                      int ops) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const float *P = poses + 16 * (size_t)b;  // block-uniform -> scalar loads
  const int src = subset ? subset[j] : j;
  const float px = tpts[3 * (size_t)src + 0], py = tpts[3 * (size_t)src + 1], pz = tpts[3 * (size_t)src + 2];
  float ox, oy, oz;
  mpx_project(P, px, py, pz, ox, oy, oz);
  float *o = out + (int64_t)b * obs + (int64_t)j * ops;
  o[0] = ox;
  o[1] = oy;
  o[2] = oz;
}

MPX_EXPORT int mpx_pose_cloud(const float *poses, int B, const float *table_pts, const int32_t *subset,
                              int n_out, float *out, int64_t out_batch_stride, int out_point_stride,
                              mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && n_out >= 0, "mpx_pose_cloud: bad size");
  if (B == 0 || n_out == 0) return 0;
  if (B > MPX_GRID_Y) {  // more poses than one launch's gridDim.y: slabs
    for (int64_t b0 = 0; b0 < B; b0 += MPX_GRID_Y)
      if (int rc = mpx_pose_cloud(poses + b0 * 16, (int)(B - b0 < MPX_GRID_Y ? B - b0 : MPX_GRID_Y), table_pts, subset, n_out,
                                  out + b0 * out_batch_stride, out_batch_stride, out_point_stride, stream))
        return rc;
    return 0;
  }
  hipLaunchKernelGGL(pose_cloud_kernel, dim3(cdiv(n_out, 256), B), dim3(256), 0, mpx_s(stream), poses,
                     table_pts, subset, n_out, out, out_batch_stride, out_point_stride);
  MPX_LAUNCH_CHECK("mpx_pose_cloud");
}

// ---- fused FK + sphere-vs-primitive SDF + collision flags ------------------------------------------
// One workgroup = COL_PPB consecutive (env, waypoint) pairs g = b*T + t.  FK for all of them runs
// on COL_PPB lanes of wave 0; then each of the 4 waves takes pairs round-robin with the S spheres
// on its lanes.  Primitive data of the pair's environment is wave-uniform (scalar loads).
// Algorithmic bytes per pair: 28 (q) + per-env primitive data amortised over T; writes 4*S when
// min_sdf is requested, else nothing but the rare atomicOr.
constexpr int COL_PPB = 16;

__global__ void __launch_bounds__(256)
    franka_collision_kernel(const float *__restrict__ q, int G, int T, float finger,
                            const float *__restrict__ sc, const float *__restrict__ sr,
                            const int32_t *__restrict__ sl, int S, const float *__restrict__ cub_f,
                            const float *__restrict__ cub_d, int M1, const float *__restrict__ cyl_f,
                            const float *__restrict__ cyl_r, const float *__restrict__ cyl_h, int M2,
                            int32_t *__restrict__ flags, float *__restrict__ min_sdf) {
  __shared__ float lds[COL_PPB * FRAME_FLOATS];
  const int g0 = blockIdx.x * COL_PPB;
  const int ng = min(COL_PPB, G - g0);
  if (threadIdx.x < ng) {
    float qq[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qq[j] = q[(size_t)(g0 + threadIdx.x) * 7 + j];
    franka_fk_frames(qq, finger, lds + threadIdx.x * FRAME_FLOATS);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = wave; i < ng; i += 4) {
    const int g = g0 + i;
    const int b = g / T;  // wave-uniform
    const float *cf = cub_f + (size_t)b * M1 * 16;
    const float *cd = cub_d + (size_t)b * M1 * 3;
    const float *yf = cyl_f + (size_t)b * M2 * 16;
    const float *yr = cyl_r + (size_t)b * M2;
    const float *yh = cyl_h + (size_t)b * M2;
    bool any_hit = false;
    for (int s = lane; s < S; s += 64) {
      float x, y, z;
      rigid_apply(lds + i * FRAME_FLOATS + 12 * sl[s], sc[3 * s + 0], sc[3 * s + 1], sc[3 * s + 2], x, y, z);
      float best = __builtin_inff();
      for (int m = 0; m < M1; ++m) {
        float v = cuboid_sdf(cf + 16 * m, cd[3 * m + 0], cd[3 * m + 1], cd[3 * m + 2], x, y, z);
        best = v < best ? v : best;
      }
      float besty = __builtin_inff();
      for (int m = 0; m < M2; ++m) {
        float v = cylinder_sdf(yf + 16 * m, yr[m], yh[m], x, y, z);
        besty = v < besty ? v : besty;
      }
      best = fminf(best, besty);  // torch.minimum(cuboids, cylinders), model.py:304-307
      if (min_sdf) min_sdf[(size_t)g * S + s] = best;
      any_hit |= best <= sr[s];  // model.py:309-311
    }
    if (__any(any_hit) && lane == 0) atomicOr(flags + b, 1);
  }
}

// Per-environment form (the default whenever M1, M2 <= 64): a workgroup owns ONE environment and a chunk of up to
// COL_TC of its waypoints, so everything that depends on the environment alone is done once:
//   * the zero-volume masks of its primitives become two 64-bit wave-uniform words (a ballot over lanes = primitives);
//     the evaluation loops walk the SET bits only (s_ff1 / clear lowest bit), and an unmasked primitive's frame and
//     sizes arrive through scalar loads as SGPR operands -- a masked row costs nothing, not even a compare;
//   * FK runs once per waypoint (lanes of the first wave), frames parked in LDS;
//   * the (waypoint, sphere) pairs of the chunk are FLATTENED over the lanes: 50 x 56 pairs fill 43.75 waves instead
//     of 50 waves that each idle 8 of 64 lanes.
// Arithmetic per (sphere, primitive) and the order of the minima are those of franka_collision_kernel: bit-identical.
template <int BLOCK, int COL_TC>
__global__ void __launch_bounds__(BLOCK)
    franka_collision_env_kernel(const float *__restrict__ q, int T, int chunks, float finger, const float *__restrict__ sc,
                                const float *__restrict__ sr, const int32_t *__restrict__ sl, int S,
                                const float *__restrict__ cub_f, const float *__restrict__ cub_d, int M1,
                                const float *__restrict__ cyl_f, const float *__restrict__ cyl_r,
                                const float *__restrict__ cyl_h, int M2, int32_t *__restrict__ flags,
                                float *__restrict__ min_sdf) {
  extern __shared__ float lds[];  // nt x FRAME_FLOATS
  const int b = blockIdx.x / chunks, t0 = (blockIdx.x - b * chunks) * COL_TC;  // (block-uniform)
  const int nt = min(COL_TC, T - t0);
  const int lane = threadIdx.x & 63;
  if ((int)threadIdx.x < nt) {
    float qq[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qq[j] = q[((size_t)b * T + t0 + threadIdx.x) * 7 + j];
    franka_fk_frames(qq, finger, lds + threadIdx.x * FRAME_FLOATS);
  }
  const float *cf = cub_f + (size_t)b * M1 * 16;
  const float *cd = cub_d + (size_t)b * M1 * 3;
  const float *yf = cyl_f + (size_t)b * M2 * 16;
  const float *yr = cyl_r + (size_t)b * M2;
  const float *yh = cyl_h + (size_t)b * M2;
  // live-primitive masks: lane m tests primitive m (every wave computes the same two words)
  bool clive = false, ylive = false;
  if (lane < M1) clive = !(mpx_is_zero(cd[3 * lane + 0]) || mpx_is_zero(cd[3 * lane + 1]) || mpx_is_zero(cd[3 * lane + 2]));
  if (lane < M2) ylive = !(mpx_is_zero(yr[lane]) || mpx_is_zero(yh[lane]));
  const unsigned long long cmask = __builtin_amdgcn_ballot_w64(clive), ymask = __builtin_amdgcn_ballot_w64(ylive);
  __syncthreads();
  const int npairs = nt * S;
  const int dq = BLOCK / S, dr = BLOCK - dq * S;  // a step of BLOCK pairs = dq waypoints + dr spheres
  int t = (int)threadIdx.x / S, s = (int)threadIdx.x - t * S;
  bool any_hit = false;
  for (int p0 = 0; p0 < npairs; p0 += BLOCK) {  // (block-uniform trip count)
    const bool on = p0 + (int)threadIdx.x < npairs;
    const int tt = on ? t : 0, ss = on ? s : 0;
    float x, y, z;
    rigid_apply(lds + tt * FRAME_FLOATS + 12 * sl[ss], sc[3 * ss + 0], sc[3 * ss + 1], sc[3 * ss + 2], x, y, z);
    float best = __builtin_inff();
    for (unsigned long long m = cmask; m; m &= m - 1) {
      const int i = __builtin_ctzll(m);  // wave-uniform: the frame and the sizes are scalar loads
      const float v = cuboid_sdf_live(cf + 16 * i, cd[3 * i + 0], cd[3 * i + 1], cd[3 * i + 2], x, y, z);
      best = v < best ? v : best;
    }
    float besty = __builtin_inff();
    for (unsigned long long m = ymask; m; m &= m - 1) {
      const int i = __builtin_ctzll(m);
      const float v = cylinder_sdf_live(yf + 16 * i, yr[i], yh[i], x, y, z);
      besty = v < besty ? v : besty;
    }
    best = fminf(best, besty);  // torch.minimum(cuboids, cylinders), model.py:304-307
    if (on) {
      if (min_sdf) min_sdf[((size_t)b * T + t0 + tt) * S + ss] = best;
      any_hit |= best <= sr[ss];  // model.py:309-311
    }
    t += dq, s += dr;
    if (s >= S) s -= S, ++t;
  }
  if (__any(any_hit) && lane == 0) atomicOr(flags + b, 1);
}

MPX_EXPORT int mpx_franka_collision(const float *q, int B, int T, float finger, const float *sph_centers,
                                    const float *sph_radii, const int32_t *sph_link, int S,
                                    const float *cub_frames, const float *cub_dims, int M1,
                                    const float *cyl_frames, const float *cyl_radii,
                                    const float *cyl_heights, int M2, int32_t *flags, float *min_sdf,
                                    mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && T >= 0 && S >= 0 && M1 >= 0 && M2 >= 0, "mpx_franka_collision: negative size");
  MPX_REQUIRE((int64_t)B * T < (int64_t)1 << 31, "mpx_franka_collision: B*T overflows int32");
  if (B == 0 || T == 0 || S == 0) return 0;
  if (M1 <= 64 && M2 <= 64 && S <= 64) {  // per-environment form: masks in two scalar words, pairs flattened over the lanes
#define COL_ENV(BLOCK, TC)                                                                                             \
  do {                                                                                                                 \
    const int chunks = cdiv(T, TC);                                                                                    \
    MPX_REQUIRE((int64_t)B * chunks < (int64_t)1 << 31, "mpx_franka_collision: too many workgroups");                  \
    hipLaunchKernelGGL((franka_collision_env_kernel<BLOCK, TC>), dim3((unsigned)(B * chunks)), dim3(BLOCK),            \
                       (size_t)min(T, TC) * FRAME_FLOATS * sizeof(float), mpx_s(stream), q, T, chunks, finger,         \
                       sph_centers, sph_radii, sph_link, S, cub_frames, cub_dims, M1, cyl_frames, cyl_radii,           \
                       cyl_heights, M2, flags, min_sdf);                                                               \
  } while (0)
    // (waypoints per workgroup, measured at 8192 x 50: 64 -> 0.486 ms, 32 -> 0.508, 16 -> 0.570; one- and two-wave
    // workgroups with 16 / 32 waypoints 0.735 / 0.673: FK runs once per chunk on as many lanes as the chunk has waypoints)
    if (T * S <= 64) COL_ENV(64, 64);  // one waypoint (the rollout step): one wave per environment
    else COL_ENV(256, 64);
#undef COL_ENV
    MPX_LAUNCH_CHECK("mpx_franka_collision");
  }
  const int G = B * T;
  hipLaunchKernelGGL(franka_collision_kernel, dim3(cdiv(G, COL_PPB)), dim3(256), 0, mpx_s(stream), q, G, T,
                     finger, sph_centers, sph_radii, sph_link, S, cub_frames, cub_dims, M1, cyl_frames,
                     cyl_radii, cyl_heights, M2, flags, min_sdf);
  MPX_LAUNCH_CHECK("mpx_franka_collision");
}

// ---- per-call robot-point subset ------------------------------------------------------------------------------------
// robofin's FrankaSampler.sample redraws np.random.choice(P, num_points, replace=False) on EVERY call, one subset for
// the whole batch (mpinets/model.py:170-181, run_inference.py:188-189).  Device form: row i of the point table gets a
// Philox key (counter (i >> 2, draw, STREAM_SUBSET, 0), key = seed); the n_out smallest (key, row) pairs are the subset,
// in key order (uniform without replacement, uniform order).  One workgroup: radix select + counting sort in LDS
// (select_device.h, shared with the scene and depth draws).  Restated in oracle/mpn_oracle.c orc_draw_subset.
enum { STREAM_SUBSET = 11 };
__global__ void __launch_bounds__(SEL_THREADS)
    draw_subset_kernel(int total, int n_out, uint32_t k0, uint32_t k1, uint32_t draw, int32_t *__restrict__ out) {
  __shared__ unsigned long long sel[SEL_CAP];
  __shared__ int hist[2048];
  __shared__ int s3[3];
  mpx_select_smallest(
      total, n_out,
      [&](int g, uint32_t (&key)[4], bool (&valid)[4]) {
        const Philox r = philox4x32((uint32_t)g, draw, STREAM_SUBSET, 0u, k0, k1);
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = r.c[u], valid[u] = 4 * g + u < total;
      },
      sel, hist, s3);
  for (int i = threadIdx.x; i < n_out; i += SEL_THREADS) out[i] = (int32_t)(uint32_t)sel[i];
}

MPX_EXPORT int mpx_draw_subset(int total, int n_out, uint64_t seed, int draw, int32_t *out, mpx_stream_t stream) {
  MPX_REQUIRE(draw >= 0, "mpx_draw_subset: negative draw index");
  MPX_REQUIRE(total >= 1 && n_out >= 1 && n_out <= total, "mpx_draw_subset: need 1 <= n_out <= total (%d of %d)", n_out, total);
  MPX_REQUIRE(n_out <= SEL_MAX_OUT, "mpx_draw_subset: n_out must be <= %d", SEL_MAX_OUT);
  MPX_REQUIRE(out, "mpx_draw_subset: NULL output");
  hipLaunchKernelGGL(draw_subset_kernel, dim3(1), dim3(SEL_THREADS), 0, mpx_s(stream), total, n_out, (uint32_t)seed,
                     (uint32_t)(seed >> 32), (uint32_t)draw, out);
  MPX_LAUNCH_CHECK("mpx_draw_subset");
}

// ---- rollout joint update ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    joint_step_kernel(const float *__restrict__ qn, const float *__restrict__ dq,
                      const float *__restrict__ limits, int n, float *__restrict__ qn_out,
                      float *__restrict__ q_out, const int32_t *__restrict__ frozen) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = i % 7;
  float v = qn[i];
  // a finished rollout (run_inference.py:180-187 `break`) keeps its last configuration
  if (!(frozen && frozen[i / 7])) v = v + dq[i];
  v = fminf(fmaxf(v, -1.0f), 1.0f);  // torch.clamp(q + self(xyz, q), min=-1, max=1), model.py:171
  const float lo = limits[2 * j], hi = limits[2 * j + 1];
  // utils.py:207-209 with limits=(-1,1): (x - (-1)) * range / 2 + lower
  const float u = (v - (-1.0f)) * (hi - lo) / 2.0f + lo;
  if (qn_out) qn_out[i] = v;
  if (q_out) q_out[i] = u;
}

MPX_EXPORT int mpx_joint_step(const float *q_norm, const float *dq, const float *limits, int B,
                              float *q_norm_out, float *q_out, const int32_t *frozen, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0, "mpx_joint_step: B < 0");
  if (B == 0) return 0;
  hipLaunchKernelGGL(joint_step_kernel, dim3(cdiv((int64_t)B * 7, 256)), dim3(256), 0, mpx_s(stream), q_norm,
                     dq, limits, B * 7, q_norm_out, q_out, frozen);
  MPX_LAUNCH_CHECK("mpx_joint_step");
}

// ---- rollout success test -------------------------------------------------------------------------
// run_inference.py:176-187: stop when the end effector (`right_gripper`) is within 1 cm and 15 deg of
// the target.  The reference does this on the host after a device->host copy every step; here it is
// a flag per environment on the device.  angle(R_eff R_t^T) < tol  <=>  (trace - 1)/2 > cos(tol).
__global__ void __launch_bounds__(64)
    franka_success_kernel(const float *__restrict__ q, const float *__restrict__ targets, int B, float finger,
                          float pos_tol, float cos_tol, int32_t *__restrict__ done, int32_t *__restrict__ steps,
                          float *__restrict__ pos_err, float *__restrict__ cos_ang) {
  __shared__ float lds[64 * FRAME_FLOATS];
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float qq[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) qq[j] = q[(size_t)b * 7 + j];
  float *fr = lds + threadIdx.x * FRAME_FLOATS;
  franka_fk_frames(qq, finger, fr);
  const float *e = fr + 12 * 14;  // right_gripper
  const float *t = targets + (size_t)b * 16;
  const float dx = e[9] - t[3], dy = e[10] - t[7], dz = e[11] - t[11];
  const float err = sqrtf(mpx_fma(dz, dz, mpx_fma(dy, dy, dx * dx)));
  float tr = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) tr = mpx_fma(e[3 * r + c], t[4 * r + c], tr);
  const float ca = (tr - 1.0f) * 0.5f;
  if (pos_err) pos_err[b] = err;
  if (cos_ang) cos_ang[b] = ca;
  const int was = done[b];
  if (steps && !was) steps[b] += 1;
  if (err < pos_tol && ca > cos_tol) done[b] = 1;
}

MPX_EXPORT int mpx_franka_success(const float *q, const float *target_poses, int B, float finger, float pos_tol,
                                  float cos_rot_tol, int32_t *done, int32_t *steps, float *pos_err,
                                  float *cos_angle, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && done != nullptr, "mpx_franka_success: bad arguments");
  if (B == 0) return 0;
  hipLaunchKernelGGL(franka_success_kernel, dim3(cdiv(B, 64)), dim3(64), 0, mpx_s(stream), q, target_poses, B,
                     finger, pos_tol, cos_rot_tol, done, steps, pos_err, cos_angle);
  MPX_LAUNCH_CHECK("mpx_franka_success");
}

// ---- batched trajectory metrics (next row N3: mpinets/metrics.py:311-384, 436-523 without PyBullet) ---------
// One wave per trajectory; lanes = waypoints (64 per pass).  Per waypoint: FK -> right_gripper pose,
// joint-limit test (metrics.py:311-322, published limits), self-collision test.  Per trajectory:
// final position error [cm] and orientation error [deg] vs the target (metrics.py:338-361), end-effector
// path lengths (metrics.py:410-434), flags.  Self collision uses the in-repo Geometric-Fabrics model
// (config/franka_fabric_config.yaml:120-140: body cylinder (0,0,-0.3)-(0,0,0.333) r 0.15 vs spheres on
// link7 (r 0.1), hand and finger tips (r 0.01)); the reference Evaluator asks PyBullet meshes instead.
// `lengths` (optional) = number of valid waypoints per trajectory (>= 1); later rows are ignored.
__device__ __forceinline__ float rot_angle_deg(const float *a, const float *b) {  // angle of A B^T, row-major 3x3
  float tr = 0.0f;
#pragma unroll
  for (int i = 0; i < 9; ++i) tr = mpx_fma(a[i], b[i], tr);
  const float c = fminf(fmaxf((tr - 1.0f) * 0.5f, -1.0f), 1.0f);
  return acosf(c) * 57.29577951308232f;
}

__global__ void __launch_bounds__(64)
    trajectory_metrics_kernel(const float *__restrict__ traj, const int32_t *__restrict__ lengths,
                              const float *__restrict__ targets, const float *__restrict__ limits, int T,
                              float finger, float *__restrict__ pos_err_cm, float *__restrict__ orient_err_deg,
                              float *__restrict__ path_pos, float *__restrict__ path_orient_deg,
                              int32_t *__restrict__ limit_violation, int32_t *__restrict__ self_collision) {
  __shared__ float frames[64 * FRAME_FLOATS];
  __shared__ float eff[65 * 12];  // right_gripper poses of this pass, slot 64 = last waypoint of the previous pass
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int len = lengths ? lengths[b] : T;
  len = len < 1 ? 1 : (len > T ? T : len);
  const float *tq = traj + (size_t)b * T * 7;
  float sum_pos = 0.0f, sum_rot = 0.0f;
  bool bad_limit = false, bad_self = false;
  for (int t0 = 0; t0 < len; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < len;
    if (valid) {
      float q[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        q[j] = tq[(size_t)t * 7 + j];
        bad_limit |= q[j] < limits[2 * j] || q[j] > limits[2 * j + 1];
      }
      float *fr = frames + lane * FRAME_FLOATS;
      franka_fk_frames(q, finger, fr);
#pragma unroll
      for (int k = 0; k < 12; ++k) eff[lane * 12 + k] = fr[12 * 14 + k];
      // self collision: distance of the sphere centres to the base segment (0,0,-0.3)-(0,0,0.333)
      const int links[4] = {7, 9, 12, 13};
      const float radii[4] = {0.1f, 0.01f, 0.01f, 0.01f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float *c = fr + 12 * links[s] + 9;
        const float zc = fminf(fmaxf(c[2], -0.3f), 0.333f);
        const float dz = c[2] - zc;
        const float d = sqrtf(mpx_fma(dz, dz, mpx_fma(c[1], c[1], c[0] * c[0])));
        bad_self |= d < 0.15f + radii[s];
      }
    }
    __syncthreads();
    if (valid && t > 0) {
      const float *cur = eff + lane * 12;
      const float *prev = lane > 0 ? eff + (lane - 1) * 12 : eff + 64 * 12;
      const float dx = cur[9] - prev[9], dy = cur[10] - prev[10], dz = cur[11] - prev[11];
      sum_pos += sqrtf(mpx_fma(dz, dz, mpx_fma(dy, dy, dx * dx)));
      sum_rot += rot_angle_deg(cur, prev);
    }
    if (valid && t == len - 1) {
      const float *cur = eff + lane * 12;
      const float *tg = targets + (size_t)b * 16;
      const float dx = cur[9] - tg[3], dy = cur[10] - tg[7], dz = cur[11] - tg[11];
      pos_err_cm[b] = 100.0f * sqrtf(mpx_fma(dz, dz, mpx_fma(dy, dy, dx * dx)));
      const float tr[9] = {tg[0], tg[1], tg[2], tg[4], tg[5], tg[6], tg[8], tg[9], tg[10]};
      orient_err_deg[b] = rot_angle_deg(cur, tr);
    }
    __syncthreads();
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 12; ++k) eff[64 * 12 + k] = eff[63 * 12 + k];
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    sum_pos += __shfl_xor(sum_pos, o);
    sum_rot += __shfl_xor(sum_rot, o);
  }
  const bool any_limit = __any(bad_limit), any_self = __any(bad_self);
  if (lane == 0) {
    path_pos[b] = sum_pos;
    path_orient_deg[b] = sum_rot;
    limit_violation[b] = any_limit;
    self_collision[b] = any_self;
  }
}

MPX_EXPORT int mpx_trajectory_metrics(const float *traj, const int32_t *lengths, const float *target_poses,
                                      const float *limits, int B, int T, float finger, float *pos_err_cm,
                                      float *orient_err_deg, float *path_pos, float *path_orient_deg,
                                      int32_t *limit_violation, int32_t *self_collision, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && T >= 1, "mpx_trajectory_metrics: bad size");
  if (B == 0) return 0;
  hipLaunchKernelGGL(trajectory_metrics_kernel, dim3(B), dim3(64), 0, mpx_s(stream), traj, lengths, target_poses,
                     limits, T, finger, pos_err_cm, orient_err_deg, path_pos, path_orient_deg, limit_violation,
                     self_collision);
  MPX_LAUNCH_CHECK("mpx_trajectory_metrics");
}
