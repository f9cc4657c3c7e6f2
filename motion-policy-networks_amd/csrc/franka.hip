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
//   * the zero-volume masks of its primitives become two 64-bit wave-uniform words (a ballot over lanes = primitives)
//     and the live primitives are compacted into LDS rows: the evaluation loops walk live rows only;
//   * FK runs once per waypoint (lanes of the first wave), frames parked in LDS;
//   * the (waypoint, sphere) pairs of the chunk are FLATTENED over the lanes (50 x 56 pairs fill 43.75 waves instead of
//     50 waves that each idle 8 of 64 lanes) and a thread keeps ALL its pairs (<= PPT sphere centres) in registers: the
//     loops run primitive-outer, pair-inner, so a primitive's frame and half sizes are fetched ONCE per wave (four
//     broadcast LDS reads of its compacted row) and feed PPT independent chains;
//   * no square root per (sphere, primitive): with d_i = |p_i| - h_i, a primitive's distance is
//     sqrt(sum max(d_i, 0)^2) + min(max_i d_i, 0), where at most one term is non-zero.  The correctly rounded sqrt is
//     monotone, so  min over primitives = sqrt(min of the sums) + min(min of the max_i d_i, 0)  EXACTLY (an inside
//     primitive zeroes the first term and makes the second the answer; with none inside the second term is 0): the
//     loop tracks two minima (2 v_min) and the sqrt runs once per sphere.  Same for cylinders (their inner rho sqrt
//     stays).  Results are bit-identical to franka_collision_kernel / oracle orc_collision_flags.
// VALU per (sphere, cuboid): projection 12 + 3 |p| - h + max3 + 3 max + 3 (sum of squares) + 2 min = 24 (was ~45 with the
// sqrt expansion, the halvings and the compare-selects inside); 18 with two pairs per packed-fp32 instruction (below).
#ifndef MPX_COL_PACK
#define MPX_COL_PACK 0  // 1: two pairs per packed-fp32 instruction (v_pk_fma_f32 ...) -- measured: no faster (see below)
#endif
#ifndef MPX_COL_TC
#define MPX_COL_TC 64  // waypoints per workgroup of the swept-sphere check (A/B: tools/ab_build.sh)
#endif
template <int BLOCK, int COL_TC, int PPT>
__global__ void __launch_bounds__(BLOCK)
    franka_collision_env_kernel(const float *__restrict__ q, int T, int chunks, float finger, const float *__restrict__ sc,
                                const float *__restrict__ sr, const int32_t *__restrict__ sl, int S,
                                const float *__restrict__ cub_f, const float *__restrict__ cub_d, int M1,
                                const float *__restrict__ cyl_f, const float *__restrict__ cyl_r,
                                const float *__restrict__ cyl_h, int M2, int32_t *__restrict__ flags,
                                float *__restrict__ min_sdf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [128 primitive rows x 16 floats | nt x FRAME_FLOATS]
  float4 *prim_rows = reinterpret_cast<float4 *>(lds);         // live cuboids from row 0, live cylinders from row 64
  float *frames = lds + 128 * 16;
  const int b = blockIdx.x / chunks, t0 = (blockIdx.x - b * chunks) * COL_TC;  // (block-uniform)
  const int nt = min(COL_TC, T - t0);
  const int lane = threadIdx.x & 63;
  if ((int)threadIdx.x < nt) {
    float qq[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) qq[j] = q[((size_t)b * T + t0 + threadIdx.x) * 7 + j];
    franka_fk_frames(qq, finger, frames + threadIdx.x * FRAME_FLOATS);
  }
  // live primitives, COMPACTED into LDS rows [R (3 x 4 floats: rotation row | Rt) | half sizes]: lane m of the staging
  // wave tests primitive m, a live one takes the slot = number of live ones before it.  The evaluation loops below read
  // a row with four broadcast ds_read_b128 (every lane the same address) -- a masked primitive costs nothing at all.
  // Staging runs on the LAST waves of the workgroup (the first one is busy with FK).
  const float *cf = cub_f + (size_t)b * M1 * 16;
  const float *cd = cub_d + (size_t)b * M1 * 3;
  const float *yf = cyl_f + (size_t)b * M2 * 16;
  const float *yr = cyl_r + (size_t)b * M2;
  const float *yh = cyl_h + (size_t)b * M2;
  bool clive = false, ylive = false;
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, r0 = 0.0f, h0 = 0.0f;
  if (lane < M1) {
    c0 = cd[3 * lane + 0], c1 = cd[3 * lane + 1], c2 = cd[3 * lane + 2];
    clive = !(mpx_is_zero(c0) || mpx_is_zero(c1) || mpx_is_zero(c2));
  }
  if (lane < M2) {
    r0 = yr[lane], h0 = yh[lane];
    ylive = !(mpx_is_zero(r0) || mpx_is_zero(h0));
  }
  const unsigned long long cmask = __builtin_amdgcn_ballot_w64(clive), ymask = __builtin_amdgcn_ballot_w64(ylive);
  const int n_cub = __builtin_popcountll(cmask), n_cyl = __builtin_popcountll(ymask);  // (every wave: the same counts)
  const unsigned long long below = ((unsigned long long)1 << lane) - 1;
  constexpr int CUB_WAVE = BLOCK / 64 - 1, CYL_WAVE = BLOCK >= 128 ? BLOCK / 64 - 2 : 0;
  const int wave = (int)threadIdx.x >> 6;
  if (wave == CUB_WAVE && clive) {
    float4 *dst = prim_rows + 4 * __builtin_popcountll(cmask & below);
    const float4 *src = reinterpret_cast<const float4 *>(cf + 16 * lane);
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2];
    dst[3] = make_float4(c0 / 2.0f, c1 / 2.0f, c2 / 2.0f, 0.0f);  // geometry.py:276
  }
  if (wave == CYL_WAVE && ylive) {
    float4 *dst = prim_rows + 4 * (64 + __builtin_popcountll(ymask & below));
    const float4 *src = reinterpret_cast<const float4 *>(yf + 16 * lane);
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2];
    dst[3] = make_float4(r0, h0 / 2.0f, 0.0f, 0.0f);  // geometry.py:489
  }
  __syncthreads();
  const int npairs = nt * S;
  // pairs of this thread: p = threadIdx.x + k * BLOCK, k < PPT (the launcher picks PPT = the chunk's pairs / BLOCK, rounded
  // up to even: straight-line code, no per-pair branches; pairs past the end repeat pair 0 and are not stored)
  // What bounds the kernel: a wave64 VALU instruction occupies its SIMD for 4 cycles, and SQ_ACTIVE_INST_VALU x 4 is
  // 98 % of the kernel's cycles (profiles/r04_collision_pmc_pass1.csv) -- VALU ISSUE, so only fewer instructions help.
  // The vector type below can hold TWO pairs per register pair (MPX_COL_PACK=1: the projection, the subtraction and the
  // sum of squares become v_pk_mul / v_pk_fma / v_pk_add_f32, 18 instead of 24 instructions per (sphere, cuboid), same
  // IEEE results per element): measured 0.322 vs 0.316 ms -- a packed fp32 instruction costs two issue slots' worth of
  // time on this chip, so the default stays one pair per instruction (profiles/r04_other_measurements.md).
  constexpr int W = (MPX_COL_PACK && PPT % 2 == 0) ? 2 : 1, NV = PPT / W;
  typedef float vec __attribute__((ext_vector_type(W)));
  vec x[NV], y[NV], z[NV], ssc[NV], dc[NV], ssy[NV], dy[NV];
  {
    const int dq = BLOCK / S, dr = BLOCK - dq * S;  // a step of BLOCK pairs = dq waypoints + dr spheres
    int tt = (int)threadIdx.x / S, ss = (int)threadIdx.x - tt * S;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      ssc[k / W][k % W] = dc[k / W][k % W] = ssy[k / W][k % W] = dy[k / W][k % W] = __builtin_inff();
      x[k / W][k % W] = y[k / W][k % W] = z[k / W][k % W] = 0.0f;
      {
        const bool on = (int)threadIdx.x + k * BLOCK < npairs;
        const int t1 = on ? tt : 0, s1 = on ? ss : 0;
        float ox, oy, oz;
        rigid_apply(frames + t1 * FRAME_FLOATS + 12 * sl[s1], sc[3 * s1 + 0], sc[3 * s1 + 1], sc[3 * s1 + 2], ox, oy, oz);
        x[k / W][k % W] = ox, y[k / W][k % W] = oy, z[k / W][k % W] = oz;
      }
      tt += dq, ss += dr;
      if (ss >= S) ss -= S, ++tt;
    }
  }
  auto vfma = [](vec a, vec b, vec c) __attribute__((always_inline)) { return __builtin_elementwise_fma(a, b, c); };
  auto vabs = [](vec a) __attribute__((always_inline)) { return __builtin_elementwise_abs(a); };
  auto vmax = [](vec a, vec b) __attribute__((always_inline)) { return __builtin_elementwise_max(a, b); };
  auto vmin = [](vec a, vec b) __attribute__((always_inline)) { return __builtin_elementwise_min(a, b); };
  // mpx_project (sdf_device.h), element-wise on the packed pairs: the same operations in the same order
  struct Frame {
    vec r[12];
  };
  auto splat = [](const float4 &f0, const float4 &f1, const float4 &f2) __attribute__((always_inline)) {
    Frame f;
    f.r[0] = (vec)f0.x, f.r[1] = (vec)f0.y, f.r[2] = (vec)f0.z, f.r[3] = (vec)f0.w;
    f.r[4] = (vec)f1.x, f.r[5] = (vec)f1.y, f.r[6] = (vec)f1.z, f.r[7] = (vec)f1.w;
    f.r[8] = (vec)f2.x, f.r[9] = (vec)f2.y, f.r[10] = (vec)f2.z, f.r[11] = (vec)f2.w;
    return f;
  };
  auto project = [&](const Frame &f, vec vx, vec vy, vec vz, vec &px, vec &py, vec &pz) __attribute__((always_inline)) {
    px = vfma(f.r[2], vz, vfma(f.r[1], vy, f.r[0] * vx)) + f.r[3];
    py = vfma(f.r[6], vz, vfma(f.r[5], vy, f.r[4] * vx)) + f.r[7];
    pz = vfma(f.r[10], vz, vfma(f.r[9], vy, f.r[8] * vx)) + f.r[11];
  };
  const vec zero = (vec)0.0f;
  for (int i = 0; i < n_cub; ++i) {
    const float4 hs = prim_rows[4 * i + 3];
    const Frame f = splat(prim_rows[4 * i + 0], prim_rows[4 * i + 1], prim_rows[4 * i + 2]);
    const vec hx = (vec)hs.x, hy = (vec)hs.y, hz = (vec)hs.z;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        vec px, py, pz;
        project(f, x[v], y[v], z[v], px, py, pz);
        const vec d0 = vabs(px) - hx, d1 = vabs(py) - hy, d2 = vabs(pz) - hz;
        const vec m0 = vmax(d0, zero), m1 = vmax(d1, zero), m2 = vmax(d2, zero);
        ssc[v] = vmin(ssc[v], vfma(m2, m2, vfma(m1, m1, m0 * m0)));
        dc[v] = vmin(dc[v], vmax(d0, vmax(d1, d2)));
      }
  }
  for (int i = 0; i < n_cyl; ++i) {
    const float4 hs = prim_rows[4 * (64 + i) + 3];
    const Frame f = splat(prim_rows[4 * (64 + i) + 0], prim_rows[4 * (64 + i) + 1], prim_rows[4 * (64 + i) + 2]);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        vec px, py, pz, rho;
        project(f, x[v], y[v], z[v], px, py, pz);
        const vec r2 = vfma(py, py, px * px);
#pragma unroll
        for (int e = 0; e < W; ++e) rho[e] = sqrtf(r2[e]);
        const vec d0 = vabs(rho) - (vec)hs.x, d1 = vabs(pz) - (vec)hs.y;
        const vec m0 = vmax(d0, zero), m1 = vmax(d1, zero);
        ssy[v] = vmin(ssy[v], vfma(m1, m1, m0 * m0));
        dy[v] = vmin(dy[v], vmax(d0, d1));
      }
  }
  bool any_hit = false;
  bool exact = min_sdf != nullptr;  // (block-uniform)
  if (!exact) {
    // Flags only (the validation sweep): `sqrt(ss) + min(d, 0) <= r` without the square root.  With d <= 0 the value IS
    // d; otherwise it is sqrt(ss), and for r >= 0  ss < r^2 (1 - 2e-6) => sqrt(ss) < r,  ss > r^2 (1 + 2e-6) => sqrt(ss) > r
    // (the margins are 16 x the rounding of the products and of a correctly rounded sqrt).  A lane whose ss falls
    // BETWEEN the two bounds (or whose radius is negative) sends its wave through the exact branch below -- same flags
    // either way, the bounds only decide which arithmetic produces them.
    const int dq = BLOCK / S, dr = BLOCK - dq * S;
    int tt = (int)threadIdx.x / S, ss = (int)threadIdx.x - tt * S;
    bool unsure = false;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      if ((int)threadIdx.x + k * BLOCK < npairs) {
        const float r = sr[ss], r2 = r * r, lo = r2 * 0.999998f, hi = r2 * 1.000002f;
        const float sc_ = ssc[k / W][k % W], dc_ = dc[k / W][k % W], sy_ = ssy[k / W][k % W], dy_ = dy[k / W][k % W];
        const bool cin = dc_ <= 0.0f, yin = dy_ <= 0.0f;
        any_hit |= (cin ? dc_ <= r : sc_ < lo) | (yin ? dy_ <= r : sy_ < lo);
        unsure |= (r < 0.0f) | (!cin & (sc_ >= lo) & (sc_ <= hi)) | (!yin & (sy_ >= lo) & (sy_ <= hi));
      }
      ss += dr, tt += dq;
      if (ss >= S) ss -= S, ++tt;
    }
    exact = __any(unsure);
    if (exact) any_hit = false;
  }
  if (exact) {
    const int dq = BLOCK / S, dr = BLOCK - dq * S;
    int tt = (int)threadIdx.x / S, ss = (int)threadIdx.x - tt * S;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      if ((int)threadIdx.x + k * BLOCK < npairs) {
        const float cub = sqrtf(ssc[k / W][k % W]) + fminf(dc[k / W][k % W], 0.0f);  // = min over the live cuboids (none: +inf)
        // (no live cylinder in this environment -- block-uniform, two environments in three: +inf without the sqrt)
        const float cyl = n_cyl ? sqrtf(ssy[k / W][k % W]) + fminf(dy[k / W][k % W], 0.0f) : __builtin_inff();
        const float best = fminf(cub, cyl);  // torch.minimum(cuboids, cylinders), model.py:304-307
        if (min_sdf) min_sdf[((size_t)b * T + t0 + tt) * S + ss] = best;
        any_hit |= best <= sr[ss];  // model.py:309-311
      }
      tt += dq, ss += dr;
      if (ss >= S) ss -= S, ++tt;
    }
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
  // (the per-environment kernel stages the 4x4 frames as float4 rows: a frame pointer that is not 16-byte aligned -- a
  // tensor view with an odd storage offset -- takes the general kernel below, which reads them with scalar loads)
  const bool frames_aligned = (((uintptr_t)cub_frames | (uintptr_t)cyl_frames) & 15) == 0;
  static_assert(MPX_COL_TC >= 1 && MPX_COL_TC <= 64 && MPX_COL_TC * 64 <= 16 * 256,
                "MPX_COL_TC: FK runs on one thread per waypoint of a chunk and a thread holds at most 16 (waypoint, sphere) pairs");
  if (M1 <= 64 && M2 <= 64 && S <= 64 && frames_aligned) {  // per-environment form: masks in two scalar words, pairs flattened over the lanes
#define COL_ENV(BLOCK, TC, PPT)                                                                                        \
  do {                                                                                                                 \
    const int chunks = cdiv(T, TC);                                                                                    \
    MPX_REQUIRE((int64_t)B * chunks < (int64_t)1 << 31, "mpx_franka_collision: too many workgroups");                  \
    hipLaunchKernelGGL((franka_collision_env_kernel<BLOCK, TC, PPT>),                                                  \
                       dim3((unsigned)(B * chunks)), dim3(BLOCK),                                                      \
                       (size_t)(128 * 16 + min(T, TC) * FRAME_FLOATS) * sizeof(float), mpx_s(stream), q, T, chunks, finger,         \
                       sph_centers, sph_radii, sph_link, S, cub_frames, cub_dims, M1, cyl_frames, cyl_radii,           \
                       cyl_heights, M2, flags, min_sdf);                                                               \
  } while (0)
    // (waypoints per workgroup, measured at 8192 x 50: 64 -> 0.486 ms, 32 -> 0.508, 16 -> 0.570; one- and two-wave
    // workgroups with 16 / 32 waypoints 0.735 / 0.673: FK runs once per chunk on as many lanes as the chunk has waypoints)
    // (round 4, pairs in registers: MPX_COL_TC waypoints x <= 64 spheres over 256 threads = MPX_COL_TC / 4 pairs per thread)
    if (T * S <= 64) COL_ENV(64, 64, 1);  // one waypoint (the rollout step): one wave per environment, one pair per lane
    else {
      switch ((min(T, MPX_COL_TC) * S + 511) / 512) {  // packed pairs per thread of a full chunk
        case 1: COL_ENV(256, MPX_COL_TC, 2); break;
        case 2: COL_ENV(256, MPX_COL_TC, 4); break;
        case 3: COL_ENV(256, MPX_COL_TC, 6); break;
        case 4: COL_ENV(256, MPX_COL_TC, 8); break;
        case 5: COL_ENV(256, MPX_COL_TC, 10); break;
        case 6: COL_ENV(256, MPX_COL_TC, 12); break;
        case 7: COL_ENV(256, MPX_COL_TC, 14); break;
        default: COL_ENV(256, MPX_COL_TC, 16); break;
      }
    }
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
