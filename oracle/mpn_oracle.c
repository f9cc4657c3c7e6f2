/*
 * oracle/mpn_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar arithmetic; the loops over independent environments are
 * OpenMP-parallel so that bench.py's cpu_baseline can use every host core -- orc_set_threads;
 * one thread gives the same bits) of the hot path of
 * NVlabs/motion-policy-networks.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.  The product
 * (motion-policy-networks_amd/) never links, imports or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_*_frames / orc_*_sdf: PINNED against golden vectors generated in the build
 *     container by importing the reference's own mpinets/geometry.py
 *     (tests/golden/gen_geometry_golden.py -> tests/golden/geometry_*.npz).
 *   - orc_fps / orc_ball_query / orc_group_points / orc_gather_points: restate the
 *     published algorithm of pointnet2_ops v3.2.0 (github.com/fishbotics/pointnet2_ops,
 *     pinned at /root/reference/docker/Dockerfile:152; call sites
 *     /root/reference/mpinets/model.py:27,366-383).  The dependency is absent from
 *     /root/reference and CUDA-only: PARITY UNPINNED.
 *   - orc_franka_*: restates robofin v0.0.1's batched URDF FK (Dockerfile:153; call sites
 *     mpinets/model.py:250,267-271,300) from the public Franka Panda URDF constants:
 *     PARITY UNPINNED.
 *
 * Floating point: every function evaluates in IEEE binary32 with the operation order
 * written here; fused multiply-adds appear only where fmaf() is written (build with
 * -ffp-contract=off).  The HIP kernels follow the same order, so index outputs can be
 * compared bit-for-bit and float outputs to ~1 ulp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* Environments never interact, so the loops over them (and over their points / queries) run in parallel; every output
 * element is computed by exactly one iteration with the same scalar arithmetic: the results do not depend on the
 * thread count.  Built without -fopenmp the pragmas vanish. */
#ifdef _OPENMP
#include <omp.h>
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PARALLEL_FOR ORC_PRAGMA(omp parallel for schedule(static))
#define ORC_PARALLEL_FOR2 ORC_PRAGMA(omp parallel for collapse(2) schedule(static))
#define ORC_PARALLEL ORC_PRAGMA(omp parallel)
#define ORC_FOR ORC_PRAGMA(omp for schedule(dynamic, 1))
ORC_API int orc_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}
#else
#define ORC_PARALLEL_FOR
#define ORC_PARALLEL_FOR2
#define ORC_PARALLEL
#define ORC_FOR
ORC_API int orc_set_threads(int n) { (void)n; return 1; }
#endif

/* ------------------------------------------------------------------------------------------
 * Primitive inverse frames.  Reference: mpinets/geometry.py:151 (quaternion normalisation),
 * :177-223 (TorchCuboids._init_frames), :409-454 (TorchCylinders._init_frames).
 * The matrix is reproduced AS WRITTEN in the reference, including `yz - wx` in both R[1][2]
 * and R[2][1] (geometry.py:212-213, :443-444).
 * frames: n x 12 floats = R (row-major 3x3) followed by Rt (3).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_prim_frames(const float *centers, const float *quats, int n, float *frames) {
  for (int i = 0; i < n; ++i) {
    const float *q = quats + 4 * i;
    const float *c = centers + 3 * i;
    float nrm = sqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));
    float w = q[0] / nrm;
    float x = -(q[1] / nrm);
    float y = -(q[2] / nrm);
    float z = -(q[3] / nrm);
    float xx = 2.0f * (x * x), yy = 2.0f * (y * y), zz = 2.0f * (z * z);
    float wx = (2.0f * w) * x, wy = (2.0f * w) * y, wz = (2.0f * w) * z;
    float xy = (2.0f * x) * y, xz = (2.0f * x) * z, yz = (2.0f * y) * z;
    float R[9];
    R[0] = (1.0f - yy) - zz; R[1] = xy - wz;          R[2] = xz + wy;
    R[3] = xy + wz;          R[4] = (1.0f - xx) - zz; R[5] = yz - wx;
    R[6] = xz - wy;          R[7] = yz - wx;          R[8] = (1.0f - xx) - yy;
    float *f = frames + 12 * i;
    for (int k = 0; k < 9; ++k) f[k] = R[k];
    for (int r = 0; r < 3; ++r) {
      float acc = R[3 * r + 0] * (-c[0]);
      acc = fmaf(R[3 * r + 1], -c[1], acc);
      acc = fmaf(R[3 * r + 2], -c[2], acc);
      f[9 + r] = acc;
    }
  }
}

/* torch.isclose(x, 0) with default rtol=1e-5, atol=1e-8  ->  |x| <= 1e-8
 * (geometry.py:56, :155-157, :385-388).  Comparison done in float like torch does for
 * float32 tensors: |x| <= (float)1e-8.                                                   */
static int is_zero(float x) { return fabsf(x) <= 1e-8f; }

static void project(const float *f, const float *p, float *o) {
  for (int r = 0; r < 3; ++r) {
    float acc = f[3 * r + 0] * p[0];
    acc = fmaf(f[3 * r + 1], p[1], acc);
    acc = fmaf(f[3 * r + 2], p[2], acc);
    o[r] = acc + f[9 + r];
  }
}

/* geometry.py:276-287 (one cuboid, one projected point) */
static float cuboid_sdf1(const float *p, const float *dims) {
  float d0 = fabsf(p[0]) - dims[0] / 2.0f;
  float d1 = fabsf(p[1]) - dims[1] / 2.0f;
  float d2 = fabsf(p[2]) - dims[2] / 2.0f;
  float m0 = fmaxf(d0, 0.0f), m1 = fmaxf(d1, 0.0f), m2 = fmaxf(d2, 0.0f);
  float outside = sqrtf(fmaf(m2, m2, fmaf(m1, m1, m0 * m0)));
  float inside = fminf(fmaxf(d0, fmaxf(d1, d2)), 0.0f);
  return outside + inside;
}

/* geometry.py:486-506 (one cylinder, one projected point) */
static float cylinder_sdf1(const float *p, float radius, float height) {
  float rho = sqrtf(fmaf(p[1], p[1], p[0] * p[0]));
  float d0 = fabsf(rho) - radius;
  float d1 = fabsf(p[2]) - height / 2.0f;
  float m0 = fmaxf(d0, 0.0f), m1 = fmaxf(d1, 0.0f);
  float outside = sqrtf(fmaf(m1, m1, m0 * m0));
  float inside = fminf(fmaxf(d0, d1), 0.0f);
  return outside + inside;
}

/* TorchCuboids.sdf / sdf_sequence (geometry.py:238-288, :290-347).
 * points [B,P,3] (P = N, or T*N flattened for sdf_sequence), out [B,P].
 * Masked (zero-volume) cuboids contribute +inf; all masked -> +inf (geometry.py:251-254). */
ORC_API void orc_cuboid_sdf(int B, int M, int P, const float *centers, const float *dims,
                            const float *quats, const float *points, float *out) {
  float *frames = (float *)malloc(sizeof(float) * 12 * (size_t)(B * M > 0 ? B * M : 1));
  orc_prim_frames(centers, quats, B * M, frames);
  ORC_PARALLEL_FOR2
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < P; ++n) {
      const float *p = points + ((size_t)b * P + n) * 3;
      float best = INFINITY;
      for (int m = 0; m < M; ++m) {
        const float *d = dims + ((size_t)b * M + m) * 3;
        if (is_zero(d[0]) || is_zero(d[1]) || is_zero(d[2])) continue;
        float q[3];
        project(frames + ((size_t)b * M + m) * 12, p, q);
        float s = cuboid_sdf1(q, d);
        best = s < best ? s : best;  /* torch.min: NaN not considered here */
      }
      out[(size_t)b * P + n] = best;
    }
  free(frames);
}

/* TorchCylinders.sdf / sdf_sequence (geometry.py:456-507, :509-568). */
ORC_API void orc_cylinder_sdf(int B, int M, int P, const float *centers, const float *radii,
                              const float *heights, const float *quats, const float *points,
                              float *out) {
  float *frames = (float *)malloc(sizeof(float) * 12 * (size_t)(B * M > 0 ? B * M : 1));
  orc_prim_frames(centers, quats, B * M, frames);
  ORC_PARALLEL_FOR2
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < P; ++n) {
      const float *p = points + ((size_t)b * P + n) * 3;
      float best = INFINITY;
      for (int m = 0; m < M; ++m) {
        float r = radii[(size_t)b * M + m], h = heights[(size_t)b * M + m];
        if (is_zero(r) || is_zero(h)) continue;
        float q[3];
        project(frames + ((size_t)b * M + m) * 12, p, q);
        float s = cylinder_sdf1(q, r, h);
        best = s < best ? s : best;
      }
      out[(size_t)b * P + n] = best;
    }
  free(frames);
}

/* TorchSpheres.sdf / sdf_sequence (geometry.py:87-123). */
ORC_API void orc_sphere_sdf(int B, int M, int P, const float *centers, const float *radii,
                            const float *points, float *out) {
  ORC_PARALLEL_FOR2
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < P; ++n) {
      const float *p = points + ((size_t)b * P + n) * 3;
      float best = INFINITY;
      for (int m = 0; m < M; ++m) {
        float r = radii[(size_t)b * M + m];
        if (is_zero(r)) continue;
        const float *c = centers + ((size_t)b * M + m) * 3;
        float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
        float s = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) - r;
        best = s < best ? s : best;
      }
      out[(size_t)b * P + n] = best;
    }
}

/* ------------------------------------------------------------------------------------------
 * Franka forward kinematics (robofin v0.0.1 restated from the public URDF; PARITY UNPINNED).
 * Frames (15): link0..link8, hand, leftfinger, rightfinger, leftfingertip, rightfingertip,
 * right_gripper.  Output T[B][15][12]: R row-major 3x3 then t.
 * ---------------------------------------------------------------------------------------- */

/* sin/cos with a fixed evaluation order so that the GPU kernel can reproduce them
 * bit-for-bit: Cody-Waite reduction by pi/2 (two constants) + degree-9/10 polynomials.      */
static void orc_sincosf(float x, float *s_out, float *c_out) {
  const float TWO_OVER_PI = 0.63661977236758134308f;
  const float PIO2_HI = 1.57079625129699707031f;      /* 0x3fc90fda */
  const float PIO2_LO = 7.54978941586159635335e-08f;  /* pi/2 - PIO2_HI */
  float kf = rintf(x * TWO_OVER_PI);
  float r = fmaf(-kf, PIO2_HI, x);
  r = fmaf(-kf, PIO2_LO, r);
  float r2 = r * r;
  float ps = fmaf(r2, 2.7557314297e-06f, -1.9841270114e-04f);
  ps = fmaf(ps, r2, 8.3333337680e-03f);
  ps = fmaf(ps, r2, -1.6666667163e-01f);
  float sn = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, -2.7557314297e-07f, 2.4801587642e-05f);
  pc = fmaf(pc, r2, -1.3888889225e-03f);
  pc = fmaf(pc, r2, 4.1666667908e-02f);
  pc = fmaf(pc, r2, -0.5f);
  float cs = fmaf(pc, r2, 1.0f);
  int k = (int)kf & 3;
  float s = (k & 1) ? cs : sn;
  float c = (k & 1) ? sn : cs;
  if (k == 1 || k == 2) c = -c;
  if (k >= 2) s = -s;
  *s_out = s;
  *c_out = c;
}

ORC_API void orc_sincos(const float *x, int n, float *s, float *c) {
  for (int i = 0; i < n; ++i) orc_sincosf(x[i], s + i, c + i);
}

/* out = a o f  (3x4 rigid composition), fixed fma order */
static void compose(const float *a, const float *f, float *o) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      float acc = a[3 * r + 0] * f[0 + c];
      acc = fmaf(a[3 * r + 1], f[3 + c], acc);
      acc = fmaf(a[3 * r + 2], f[6 + c], acc);
      o[3 * r + c] = acc;
    }
    float acc = a[9 + r];
    acc = fmaf(a[3 * r + 0], f[9], acc);
    acc = fmaf(a[3 * r + 1], f[10], acc);
    acc = fmaf(a[3 * r + 2], f[11], acc);
    o[9 + r] = acc;
  }
}

/* o = p with its rotation post-multiplied by Rz(theta) */
static void rotz(const float *p, float s, float c, float *o) {
  for (int r = 0; r < 3; ++r) {
    float a = p[3 * r + 0], b = p[3 * r + 1];
    o[3 * r + 0] = fmaf(b, s, a * c);
    o[3 * r + 1] = fmaf(b, c, -(a * s));
    o[3 * r + 2] = p[3 * r + 2];
    o[9 + r] = p[9 + r];
  }
}

#define SQRT_HALF 0.70710678118654752440f
#define RX_NEG /* Rx(-pi/2) */ {1, 0, 0, 0, 0, 1, 0, -1, 0}
#define RX_POS /* Rx(+pi/2) */ {1, 0, 0, 0, 0, -1, 0, 1, 0}

ORC_API void orc_franka_fk(const float *q, int B, float finger, float *T) {
  /* joint origins of the Franka Panda URDF: {R(9), t(3)} */
  static const float J[7][12] = {
      {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.0f, 0.0f, 0.333f},
      {1, 0, 0, 0, 0, 1, 0, -1, 0, 0.0f, 0.0f, 0.0f},
      {1, 0, 0, 0, 0, -1, 0, 1, 0, 0.0f, -0.316f, 0.0f},
      {1, 0, 0, 0, 0, -1, 0, 1, 0, 0.0825f, 0.0f, 0.0f},
      {1, 0, 0, 0, 0, 1, 0, -1, 0, -0.0825f, 0.384f, 0.0f},
      {1, 0, 0, 0, 0, -1, 0, 1, 0, 0.0f, 0.0f, 0.0f},
      {1, 0, 0, 0, 0, -1, 0, 1, 0, 0.088f, 0.0f, 0.0f},
  };
  static const float F_LINK8[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.0f, 0.0f, 0.107f};
  static const float F_HAND[12] = {SQRT_HALF, SQRT_HALF, 0, -SQRT_HALF, SQRT_HALF, 0, 0, 0, 1, 0, 0, 0};
  static const float F_TIP[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.0f, 0.0f, 0.045f};
  static const float F_GRIP[12] = {-SQRT_HALF, -SQRT_HALF, 0, SQRT_HALF, -SQRT_HALF, 0, 0, 0, 1, 0.0f, 0.0f, 0.1f};
  ORC_PARALLEL_FOR
  for (int b = 0; b < B; ++b) {
    float *out = T + (size_t)b * 15 * 12;
    float cur[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    memcpy(out, cur, sizeof(cur));
    for (int j = 0; j < 7; ++j) {
      float tmp[12], s, c;
      compose(cur, J[j], tmp);
      orc_sincosf(q[(size_t)b * 7 + j], &s, &c);
      rotz(tmp, s, c, cur);
      memcpy(out + 12 * (j + 1), cur, sizeof(cur));
    }
    float *l8 = out + 12 * 8, *hand = out + 12 * 9;
    compose(cur, F_LINK8, l8);
    compose(l8, F_HAND, hand);
    float fl[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.0f, finger, 0.0584f};
    float fr[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0.0f, -finger, 0.0584f};
    compose(hand, fl, out + 12 * 10);
    compose(hand, fr, out + 12 * 11);
    compose(out + 12 * 10, F_TIP, out + 12 * 12);
    compose(out + 12 * 11, F_TIP, out + 12 * 13);
    compose(l8, F_GRIP, out + 12 * 14);
  }
}

/* rigidly move table points by their link frame: out[b,j] = T[b,link[src]] * p[src],
 * src = subset ? subset[j] : j.  Restates FrankaSampler.sample.        */
ORC_API void orc_transform_table(const float *T, int B, int n_frames, const float *pts,
                                 const int32_t *link, const int32_t *subset, int n_out,
                                 float *out) {
  ORC_PARALLEL_FOR
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < n_out; ++j) {
      int src = subset ? subset[j] : j;
      const float *f = T + ((size_t)b * n_frames + link[src]) * 12;
      const float *p = pts + 3 * (size_t)src;
      float *o = out + ((size_t)b * n_out + j) * 3;
      for (int r = 0; r < 3; ++r) {
        float acc = f[3 * r + 0] * p[0];
        acc = fmaf(f[3 * r + 1], p[1], acc);
        acc = fmaf(f[3 * r + 2], p[2], acc);
        o[r] = acc + f[9 + r];
      }
    }
}

/* Swept-sphere collision reduce, mpinets/model.py:293-314:
 *   has_collision[b] = any_{t,s} ( min(cuboid_sdf, cylinder_sdf)(sphere centre) <= radius_s )
 * centres [B,T,S,3]; min_sdf (optional) [B,T,S].                                          */
ORC_API void orc_collision_flags(int B, int T, int S, const float *centres, const float *radii,
                                 int M1, const float *cc, const float *cd, const float *cq,
                                 int M2, const float *yc, const float *yr, const float *yh,
                                 const float *yq, uint8_t *flags, float *min_sdf) {
  int P = T * S;
  float *a = (float *)malloc(sizeof(float) * (size_t)B * P);
  float *c = (float *)malloc(sizeof(float) * (size_t)B * P);
  orc_cuboid_sdf(B, M1, P, cc, cd, cq, centres, a);
  orc_cylinder_sdf(B, M2, P, yc, yr, yh, yq, centres, c);
  ORC_PARALLEL_FOR
  for (int b = 0; b < B; ++b) {
    uint8_t f = 0;
    for (int i = 0; i < P; ++i) {
      float s = fminf(a[(size_t)b * P + i], c[(size_t)b * P + i]);
      if (min_sdf) min_sdf[(size_t)b * P + i] = s;
      if (s <= radii[i % S]) f = 1;
    }
    flags[b] = f;
  }
  free(a);
  free(c);
}

/* ------------------------------------------------------------------------------------------
 * pointnet2_ops v3.2.0 restated (PARITY UNPINNED -- see header).
 * ---------------------------------------------------------------------------------------- */

/* opt_n_threads(): largest power of two <= work_size, clamped to [1, 512] */
ORC_API int orc_opt_n_threads(int work_size) {
  int p = 1;
  while (p * 2 <= work_size && p * 2 <= 512) p *= 2;
  return p;
}

/* THE squared distance of the index path (one definition; the kernels' twin is mpx_sqdist() in csrc/common.h).
 * Upstream writes `(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)` (sampling_gpu.cu, ball_query_gpu.cu) and
 * `(x2*x2) + (y2*y2) + (z2*z2)`, and nvcc contracts them (-fmad=true).  LLVM's DAG combiner (NVVM is built on it) fuses
 * the LEFT multiply of each add first: (a*a + b*b) -> fma(a,a, b*b), then (. + c*c) -> fma(c,c, .): the product rounded
 * on its own is dy^2.  Evidence: profiles/r03_contraction_evidence.md (x86 and gfx950 disassembly of exactly that
 * expression under -ffp-contract=fast).  Order 1 is what rounds 1-2 assumed (dx^2 rounded on its own); it is kept so
 * that tests/test_oracle_pointnet.py can COUNT how many clouds change between the two (the residual risk, since nvcc
 * itself cannot be run here).  Not thread-safe by design: a test sets it, runs, and sets it back.                 */
static int g_sqdist_order = 0;
ORC_API void orc_set_sqdist_order(int order) { g_sqdist_order = order; }
ORC_API int orc_get_sqdist_order(void) { return g_sqdist_order; }
static float sqdist3(float dx, float dy, float dz) {
  if (g_sqdist_order == 1) return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}
static float sqdist(const float *a, const float *b) {
  return sqdist3(a[0] - b[0], a[1] - b[1], a[2] - b[2]);
}

/* furthest_point_sampling_kernel: start at index 0; temp = 1e10; points with |p|^2 <= 1e-3
 * are skipped; each of `bs` threads scans k = tid, tid+bs, ... keeping the first strictly
 * greater value; the shared-memory tree reduce (slot t absorbs slot t+s, ties keep slot t)
 * is emulated literally -- on ties it favours the smallest bit-reversed thread id.       */
ORC_API void orc_fps(const float *xyz, int B, int N, int stride, int npoint, int32_t *idx) {
  if (npoint <= 0) return;
  int bs = orc_opt_n_threads(N);
  ORC_PARALLEL /* (environments are independent: one scratch set per thread) */
  {
  float *temp = (float *)malloc(sizeof(float) * (size_t)N);
  float *tv = (float *)malloc(sizeof(float) * (size_t)bs);
  int *ti = (int *)malloc(sizeof(int) * (size_t)bs);
  ORC_FOR
  for (int b = 0; b < B; ++b) {
    const float *pts = xyz + (size_t)b * N * stride;
    int32_t *out = idx + (size_t)b * npoint;
    for (int k = 0; k < N; ++k) temp[k] = 1e10f;
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < npoint; ++j) {
      const float *p1 = pts + (size_t)old * stride;
      for (int t = 0; t < bs; ++t) {
        int besti = 0;
        float best = -1.0f;
        for (int k = t; k < N; k += bs) {
          const float *p2 = pts + (size_t)k * stride;
          float mag = sqdist3(p2[0], p2[1], p2[2]);
          if ((double)mag <= 1e-3) continue; /* upstream: float mag vs the double literal 1e-3 */
          float d = sqdist(p2, p1);
          float d2 = fminf(d, temp[k]);
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        tv[t] = best;
        ti[t] = besti;
      }
      for (int s = bs / 2; s >= 1; s /= 2)
        for (int t = 0; t < s; ++t) {
          float v1 = tv[t], v2 = tv[t + s];
          int i1 = ti[t], i2 = ti[t + s];
          tv[t] = v1 > v2 ? v1 : v2;
          ti[t] = v2 > v1 ? i2 : i1;
        }
      old = ti[0];
      out[j] = old;
    }
  }
  free(temp);
  free(tv);
  free(ti);
  }
}

/* gather_points_kernel, on [B,N,stride] rows: out[b,j,:3] = xyz[b,idx[b,j],:3] */
ORC_API void orc_gather_points(const float *xyz, int B, int N, int stride, const int32_t *idx,
                               int npoint, float *out) {
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < npoint; ++j)
      for (int c = 0; c < 3; ++c)
        out[((size_t)b * npoint + j) * 3 + c] =
            xyz[((size_t)b * N + idx[(size_t)b * npoint + j]) * stride + c];
}

/* query_ball_point_kernel: first `nsample` indices k (ascending) with d2 < r*r; all slots
 * pre-filled with the first hit; output zero-initialised.                                */
ORC_API void orc_ball_query(const float *new_xyz, const float *xyz, int B, int N, int stride,
                            int npoint, float radius, int nsample, int32_t *idx, int32_t *cnt_out) {
  float r2 = radius * radius;
  memset(idx, 0, sizeof(int32_t) * (size_t)B * npoint * nsample);
  ORC_PARALLEL_FOR2
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < npoint; ++j) {
      const float *c = new_xyz + ((size_t)b * npoint + j) * 3;
      int32_t *o = idx + ((size_t)b * npoint + j) * nsample;
      int cnt = 0;
      for (int k = 0; k < N && cnt < nsample; ++k) {
        const float *p = xyz + ((size_t)b * N + k) * stride;
        float dx = c[0] - p[0], dy = c[1] - p[1], dz = c[2] - p[2];
        float d2 = sqdist3(dx, dy, dz);
        if (d2 < r2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
        }
      }
      if (cnt_out) cnt_out[(size_t)b * npoint + j] = cnt; /* hits found (<= nsample) */
    }
}

/* QueryAndGroup (use_xyz=True): out[b, 0:3, j, l] = xyz[idx] - new_xyz[j];
 * out[b, 3:3+C, j, l] = feat[b, :, idx].  feat is [B,C,N] (channel-major, like the
 * reference's features tensor).  out [B,3+C,npoint,nsample].                            */
ORC_API void orc_group_points(const float *xyz, int stride, const float *new_xyz,
                              const float *feat, const int32_t *idx, int B, int N, int C,
                              int npoint, int nsample, float *out) {
  size_t plane = (size_t)npoint * nsample;
  ORC_PARALLEL_FOR2
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < npoint; ++j)
      for (int l = 0; l < nsample; ++l) {
        int k = idx[((size_t)b * npoint + j) * nsample + l];
        float *o = out + (size_t)b * (3 + C) * plane + (size_t)j * nsample + l;
        for (int c = 0; c < 3; ++c)
          o[c * plane] = xyz[((size_t)b * N + k) * stride + c] -
                         new_xyz[((size_t)b * npoint + j) * 3 + c];
        for (int c = 0; c < C; ++c)
          o[(3 + c) * plane] = feat[((size_t)b * C + c) * N + k];
      }
}

/* ------------------------------------------------------------------------------------------
 * Batched scene clouds.  The reference (mpinets/geometry.py:571-608) draws from NumPy's global
 * RNG inside geometrout, so only its DISTRIBUTION is defined: pools of int(p_i*N)+500 i.i.d.
 * surface samples per obstacle (:598-604), N pool slots without replacement in random order
 * (:608), labels a shuffled 1..K (:594-595).  The engine samples that distribution with
 * Philox4x32-10 (csrc/scene.hip); this is the same procedure restated for bit-exact checks of
 * the obstacle ids / labels and ulp-level checks of the coordinates.  Distributional agreement
 * with the reference's own function is tested separately (tests/test_scene_cloud.py).
 * ---------------------------------------------------------------------------------------- */
static void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

ORC_API void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc_philox(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

static float orc_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

static double obstacle_area(int m, int M1, const float *cd, const float *yr, const float *yh) {
  const double PI = 3.14159265358979323846;
  if (m < M1) {
    const float *d = cd + 3 * m;
    if (is_zero(d[0]) || is_zero(d[1]) || is_zero(d[2])) return 0.0;
    return 2.0 * ((double)d[0] * d[1] + (double)d[0] * d[2] + (double)d[1] * d[2]);
  }
  float r = yr[m - M1], h = yh[m - M1];
  if (is_zero(r) || is_zero(h)) return 0.0;
  return 2.0 * PI * (double)r * (double)h + 2.0 * PI * (double)r * (double)r;
}

static int orc_cmp_u64_fwd(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* assign uint16 [B,N]; labels uint8 [B,M1+M2]; n_obstacles int32 [B] */
ORC_API void orc_scene_assign(const float *cub_dims, int M1, const float *cyl_radii, const float *cyl_heights,
                              int M2, int B, int N, uint64_t seed, int64_t env_offset, uint16_t *assign,
                              uint8_t *labels, int32_t *n_obstacles) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), env0 = (uint32_t)env_offset;
  const int M = M1 + M2;
  uint32_t *rem = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(M > 0 ? M : 1));
  for (int b = 0; b < B; ++b) {
    const float *cd = cub_dims + (size_t)b * M1 * 3, *yr = cyl_radii + (size_t)b * M2, *yh = cyl_heights + (size_t)b * M2;
    double total = 0.0;
    int K = 0;
    for (int m = 0; m < M; ++m) {
      double a = obstacle_area(m, M1, cd, yr, yh);
      total += a;
      K += a > 0.0;
    }
    n_obstacles[b] = K;
    uint16_t *arow = assign + (size_t)b * N;
    uint8_t *lab = labels + (size_t)b * M;
    if (K == 0) {
      for (int j = 0; j < N; ++j) arow[j] = 0xFFFF;
      for (int m = 0; m < M; ++m) lab[m] = 0;
      continue;
    }
    uint32_t pool = 0;
    for (int m = 0; m < M; ++m) {
      double a = obstacle_area(m, M1, cd, yr, yh);
      rem[m] = a > 0.0 ? (uint32_t)(int)((a / total) * (double)N) + 500u : 0u;
      pool += rem[m];
    }
    int live = 0;
    for (int m = 0; m < M; ++m) lab[m] = rem[m] ? (uint8_t)(++live) : 0;
    int i = K - 1;
    uint32_t ctr = 0;
    for (int m = M - 1; m >= 0 && i > 0; --m) {
      if (!rem[m]) continue;
      uint32_t r[4];
      orc_philox(ctr++, env0 + (uint32_t)b, 2u, 0u, k0, k1, r);
      int jpos = (int)(((uint64_t)r[0] * (uint32_t)(i + 1)) >> 32);
      int t = -1;
      for (int mm = 0; mm < M; ++mm)
        if (rem[mm] && ++t == jpos) {
          uint8_t tmp = lab[m];
          lab[m] = lab[mm];
          lab[mm] = tmp;
          break;
        }
      --i;
    }
    /* N of the pool slots, uniformly without replacement, in uniform order (np.random.choice, geometry.py:608):
       every slot gets a Philox key, the N smallest (key, slot) pairs in ascending order; only the owner counts */
    {
      uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)pool);
      for (uint32_t s = 0; s < pool; ++s) {
        uint32_t r[4];
        orc_philox(s >> 2, env0 + (uint32_t)b, 1u, 0u, k0, k1, r); /* one block keys four consecutive slots */
        keys[s] = ((uint64_t)r[s & 3] << 32) | s;
      }
      qsort(keys, (size_t)pool, sizeof(uint64_t), orc_cmp_u64_fwd);
      for (int j = 0; j < N; ++j) {
        uint32_t slot = (uint32_t)keys[j], acc = 0;
        int m = 0;
        while (slot >= acc + rem[m]) acc += rem[m++];
        arow[j] = (uint16_t)m;
      }
      free(keys);
    }
  }
  free(rem);
}

/* out [B,N,3]: one uniform surface sample per assigned obstacle id */
ORC_API void orc_scene_points(const float *cub_c, const float *cub_d, const float *cub_q, int M1,
                              const float *cyl_c, const float *cyl_r, const float *cyl_h, const float *cyl_q,
                              int M2, int B, int N, uint64_t seed, int64_t env_offset, const uint16_t *assign,
                              float *out) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), env0 = (uint32_t)env_offset;
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < N; ++j) {
      float *o = out + ((size_t)b * N + j) * 3;
      int m = assign[(size_t)b * N + j];
      if (m == 0xFFFF) {
        o[0] = o[1] = o[2] = 0.0f;
        continue;
      }
      uint32_t r[4];
      orc_philox((uint32_t)j, env0 + (uint32_t)b, 3u, 0u, k0, k1, r);
      float u0 = orc_u01(r[0]), u1 = orc_u01(r[1]), u2 = orc_u01(r[2]), u3 = orc_u01(r[3]);
      float lx, ly, lz;
      const float *ctr, *q;
      if (m < M1) {
        const float *d = cub_d + ((size_t)b * M1 + m) * 3;
        float ax = d[1] * d[2], ay = d[0] * d[2], az = d[0] * d[1];
        float t = u0 * (ax + ay + az);
        float sgn = u1 < 0.5f ? -0.5f : 0.5f;
        if (t < ax) {
          lx = sgn * d[0]; ly = (u2 - 0.5f) * d[1]; lz = (u3 - 0.5f) * d[2];
        } else if (t < ax + ay) {
          lx = (u2 - 0.5f) * d[0]; ly = sgn * d[1]; lz = (u3 - 0.5f) * d[2];
        } else {
          lx = (u2 - 0.5f) * d[0]; ly = (u3 - 0.5f) * d[1]; lz = sgn * d[2];
        }
        ctr = cub_c + ((size_t)b * M1 + m) * 3;
        q = cub_q + ((size_t)b * M1 + m) * 4;
      } else {
        int c = m - M1;
        float rad = cyl_r[(size_t)b * M2 + c], h = cyl_h[(size_t)b * M2 + c];
        float side = 2.0f * rad * h, cap = rad * rad;
        float t = u0 * (side + 2.0f * cap);
        float s, co;
        orc_sincosf(u1 * 6.28318530717958647692f, &s, &co);
        float rho = rad;
        lz = (u2 - 0.5f) * h;
        if (t >= side) {
          rho = rad * sqrtf(u3);
          lz = t < side + cap ? -0.5f * h : 0.5f * h;
        }
        lx = rho * co;
        ly = rho * s;
        ctr = cyl_c + ((size_t)b * M2 + c) * 3;
        q = cyl_q + ((size_t)b * M2 + c) * 4;
      }
      float n = sqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));
      float w = q[0] / n, a = q[1] / n, bb = q[2] / n, c = q[3] / n;
      float r00 = 1.0f - 2.0f * (bb * bb + c * c), r01 = 2.0f * (a * bb - w * c), r02 = 2.0f * (a * c + w * bb);
      float r10 = 2.0f * (a * bb + w * c), r11 = 1.0f - 2.0f * (a * a + c * c), r12 = 2.0f * (bb * c - w * a);
      float r20 = 2.0f * (a * c - w * bb), r21 = 2.0f * (bb * c + w * a), r22 = 1.0f - 2.0f * (a * a + bb * bb);
      o[0] = fmaf(r02, lz, fmaf(r01, ly, r00 * lx)) + ctr[0];
      o[1] = fmaf(r12, lz, fmaf(r11, ly, r10 * lx)) + ctr[1];
      o[2] = fmaf(r22, lz, fmaf(r21, ly, r20 * lx)) + ctr[2];
    }
}

/* run_inference.py:176-187 success test on given end-effector frames (R row-major + t) */
ORC_API void orc_success(const float *eff_frames, const float *targets, int B, float pos_tol, float cos_tol,
                         uint8_t *ok, float *pos_err, float *cos_ang) {
  for (int b = 0; b < B; ++b) {
    const float *e = eff_frames + (size_t)b * 12, *t = targets + (size_t)b * 16;
    float dx = e[9] - t[3], dy = e[10] - t[7], dz = e[11] - t[11];
    float err = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    float tr = 0.0f;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) tr = fmaf(e[3 * r + c], t[4 * r + c], tr);
    float ca = (tr - 1.0f) * 0.5f;
    pos_err[b] = err;
    cos_ang[b] = ca;
    ok[b] = err < pos_tol && ca > cos_tol;
  }
}

/* ------------------------------------------------------------------------------------------
 * Batched trajectory metrics (row N3).  Restates the PyBullet-free parts of the reference's
 * Evaluator: violates_joint_limits (metrics.py:311-322), check_final_position (:338-347),
 * check_final_orientation (:349-361), calculate_eff_path_lengths (:410-434).  Self collision
 * is the engine's sphere-vs-body-cylinder model from config/franka_fabric_config.yaml:120-140
 * (the reference asks PyBullet): PARITY UNPINNED for that flag.
 * ---------------------------------------------------------------------------------------- */
static float orc_rot_angle_deg(const float *a, const float *b) {
  float tr = 0.0f;
  for (int i = 0; i < 9; ++i) tr = fmaf(a[i], b[i], tr);
  float c = fminf(fmaxf((tr - 1.0f) * 0.5f, -1.0f), 1.0f);
  return acosf(c) * 57.29577951308232f;
}

ORC_API void orc_trajectory_metrics(const float *traj, const int32_t *lengths, const float *targets,
                                    const float *limits, int B, int T, float finger, float *pos_err_cm,
                                    float *orient_err_deg, float *path_pos, float *path_orient_deg,
                                    int32_t *limit_violation, int32_t *self_collision) {
  float *T1 = (float *)malloc(sizeof(float) * 15 * 12);
  for (int b = 0; b < B; ++b) {
    int len = lengths ? lengths[b] : T;
    len = len < 1 ? 1 : (len > T ? T : len);
    float prev[12], cur[12];
    double sp = 0.0, sr = 0.0;
    int bad_l = 0, bad_s = 0;
    for (int t = 0; t < len; ++t) {
      const float *q = traj + ((size_t)b * T + t) * 7;
      for (int j = 0; j < 7; ++j) bad_l |= q[j] < limits[2 * j] || q[j] > limits[2 * j + 1];
      orc_franka_fk(q, 1, finger, T1);
      memcpy(cur, T1 + 12 * 14, sizeof(cur));
      const int links[4] = {7, 9, 12, 13};
      const float radii[4] = {0.1f, 0.01f, 0.01f, 0.01f};
      for (int s = 0; s < 4; ++s) {
        const float *c = T1 + 12 * links[s] + 9;
        float zc = fminf(fmaxf(c[2], -0.3f), 0.333f);
        float dz = c[2] - zc;
        float d = sqrtf(fmaf(dz, dz, fmaf(c[1], c[1], c[0] * c[0])));
        bad_s |= d < 0.15f + radii[s];
      }
      if (t > 0) {
        float dx = cur[9] - prev[9], dy = cur[10] - prev[10], dz = cur[11] - prev[11];
        sp += sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        sr += orc_rot_angle_deg(cur, prev);
      }
      memcpy(prev, cur, sizeof(cur));
    }
    const float *tg = targets + (size_t)b * 16;
    float dx = cur[9] - tg[3], dy = cur[10] - tg[7], dz = cur[11] - tg[11];
    pos_err_cm[b] = 100.0f * sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    float tr[9] = {tg[0], tg[1], tg[2], tg[4], tg[5], tg[6], tg[8], tg[9], tg[10]};
    orient_err_deg[b] = orc_rot_angle_deg(cur, tr);
    path_pos[b] = (float)sp;
    path_orient_deg[b] = (float)sr;
    limit_violation[b] = bad_l;
    self_collision[b] = bad_s;
  }
  free(T1);
}

/* ------------------------------------------------------------------------------------------
 * Depth-camera scene clouds (row N4).  The reference renders with PyBullet and back-projects the
 * depth image (run_inference.py:194-257; robofin Bullet.get_pointcloud_from_camera, absent):
 * PARITY UNPINNED.  This restates the engine's documented definition: analytic ray casting of the
 * primitives from an OpenGL-convention camera (x right, y up, looks along -z -- the convention of the
 * reference's evaluation poses), robot pixels (collision spheres) and misses dropped, then the n_out
 * valid pixels with the smallest Philox keys, in key order (ties by pixel id).
 * frames here are the 12-float [R | t] rows of orc_prim_frames.
 * ---------------------------------------------------------------------------------------- */
static void orc_pixel_ray(const float *P, float fx, float fy, float cx, float cy, int u, int v, float *d) {
  float xc = ((float)u + 0.5f - cx) / fx, yc = -(((float)v + 0.5f - cy) / fy), zc = -1.0f;
  float wx = P[0] * xc, wy = P[4] * xc, wz = P[8] * xc;
  wx = fmaf(P[1], yc, wx), wy = fmaf(P[5], yc, wy), wz = fmaf(P[9], yc, wz);
  wx = fmaf(P[2], zc, wx), wy = fmaf(P[6], zc, wy), wz = fmaf(P[10], zc, wz);
  float n = sqrtf(fmaf(wz, wz, fmaf(wy, wy, wx * wx)));
  d[0] = wx / n, d[1] = wy / n, d[2] = wz / n;
}
static void orc_rot12(const float *f, const float *x, float *o) {
  for (int r = 0; r < 3; ++r) o[r] = fmaf(f[3 * r + 2], x[2], fmaf(f[3 * r + 1], x[1], f[3 * r] * x[0]));
}
static void orc_proj12(const float *f, const float *x, float *o) {
  for (int r = 0; r < 3; ++r) {
    float a = f[3 * r] * x[0];
    a = fmaf(f[3 * r + 1], x[1], a);
    a = fmaf(f[3 * r + 2], x[2], a);
    o[r] = a + f[9 + r];
  }
}
static int orc_zero(float x) { return fabsf(x) <= 1e-8f; }

static float orc_ray_cuboid(const float *f, const float *h, const float *o, const float *d) {
  float lo[3], ld[3], tn = -INFINITY, tf = INFINITY;
  orc_proj12(f, o, lo);
  orc_rot12(f, d, ld);
  for (int a = 0; a < 3; ++a) {
    if (fabsf(ld[a]) < 1e-12f) {
      if (fabsf(lo[a]) > h[a]) return INFINITY;
    } else {
      float t1 = (-h[a] - lo[a]) / ld[a], t2 = (h[a] - lo[a]) / ld[a];
      tn = fmaxf(tn, fminf(t1, t2));
      tf = fminf(tf, fmaxf(t1, t2));
    }
  }
  return (tn <= tf && tn > 0.0f) ? tn : INFINITY;
}
static float orc_ray_cylinder(const float *f, float r, float hh, const float *o, const float *d) {
  float l[3], e[3], best = INFINITY;
  orc_proj12(f, o, l);
  orc_rot12(f, d, e);
  float a = fmaf(e[1], e[1], e[0] * e[0]);
  if (a > 1e-12f) {
    float b = fmaf(l[1], e[1], l[0] * e[0]), c = fmaf(l[1], l[1], l[0] * l[0]) - r * r;
    float disc = b * b - a * c;
    if (disc >= 0.0f) {
      float s = (-b - sqrtf(disc)) / a;
      if (s > 0.0f && fabsf(fmaf(s, e[2], l[2])) <= hh) best = s;
    }
  }
  if (fabsf(e[2]) > 1e-12f) {
    for (int k = 0; k < 2; ++k) {
      float s = ((k ? -hh : hh) - l[2]) / e[2];
      float px = fmaf(s, e[0], l[0]), py = fmaf(s, e[1], l[1]);
      if (s > 0.0f && s < best && fmaf(py, py, px * px) <= r * r) best = s;
    }
  }
  return best;
}
static float orc_ray_sphere(const float *c, float r, const float *o, const float *d) {
  float m[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
  float b = fmaf(m[2], d[2], fmaf(m[1], d[1], m[0] * d[0]));
  float cc = fmaf(m[2], m[2], fmaf(m[1], m[1], m[0] * m[0])) - r * r;
  float disc = b * b - cc;
  if (disc < 0.0f) return INFINITY;
  float s = -b - sqrtf(disc);
  return s > 0.0f ? s : INFINITY;
}

ORC_API void orc_depth_render(const float *cam, float fx, float fy, float cx, float cy, int W, int H, int B,
                              const float *cub_f, const float *cub_d, int M1, const float *cyl_f, const float *cyl_r,
                              const float *cyl_h, int M2, const float *sph_c, const float *sph_r, int S,
                              float far_clip, float *depth) {
  for (int b = 0; b < B; ++b) {
    const float *P = cam + 16 * (size_t)b;
    const float o[3] = {P[3], P[7], P[11]};
    for (int pix = 0; pix < W * H; ++pix) {
      float d[3], best = far_clip;
      orc_pixel_ray(P, fx, fy, cx, cy, pix % W, pix / W, d);
      for (int m = 0; m < M1; ++m) {
        size_t pm = (size_t)b * M1 + m;
        const float *dm = cub_d + 3 * pm;
        if (orc_zero(dm[0]) || orc_zero(dm[1]) || orc_zero(dm[2])) continue;
        float h[3] = {dm[0] / 2.0f, dm[1] / 2.0f, dm[2] / 2.0f};
        best = fminf(best, orc_ray_cuboid(cub_f + 12 * pm, h, o, d));
      }
      for (int m = 0; m < M2; ++m) {
        size_t pm = (size_t)b * M2 + m;
        if (orc_zero(cyl_r[pm]) || orc_zero(cyl_h[pm])) continue;
        best = fminf(best, orc_ray_cylinder(cyl_f + 12 * pm, cyl_r[pm], cyl_h[pm] / 2.0f, o, d));
      }
      int robot = 0;
      for (int s = 0; s < S; ++s)
        if (orc_ray_sphere(sph_c + ((size_t)b * S + s) * 3, sph_r[s], o, d) < best) robot = 1;
      depth[(size_t)b * W * H + pix] = (robot || !(best < far_clip)) ? -1.0f : best;
    }
  }
}

static int orc_cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* out [B, n_out, 3]; count [B] = valid pixels; an environment with fewer than n_out valid pixels is left untouched */
ORC_API void orc_depth_select(const float *depth, const float *cam, float fx, float fy, float cx, float cy, int W,
                              int H, int B, int n_out, uint32_t k0, uint32_t k1, int64_t env_offset, float *out,
                              int32_t *count) {
  const uint32_t env0 = (uint32_t)env_offset;
  uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)W * H);
  for (int b = 0; b < B; ++b) {
    const float *dp = depth + (size_t)b * W * H, *P = cam + 16 * (size_t)b;
    int n = 0;
    for (int pix = 0; pix < W * H; ++pix) {
      if (dp[pix] < 0.0f) continue;
      uint32_t r[4];
      orc_philox((uint32_t)(pix >> 2), env0 + (uint32_t)b, 9u, 0u, k0, k1, r); /* one block keys four consecutive pixels */
      keys[n++] = ((uint64_t)r[pix & 3] << 32) | (uint32_t)pix;
    }
    count[b] = n;
    if (n < n_out) continue;
    qsort(keys, (size_t)n, sizeof(uint64_t), orc_cmp_u64);
    for (int i = 0; i < n_out; ++i) {
      int pix = (int)(uint32_t)keys[i];
      float d[3];
      orc_pixel_ray(P, fx, fy, cx, cy, pix % W, pix / W, d);
      float *o = out + ((size_t)b * n_out + i) * 3;
      o[0] = fmaf(dp[pix], d[0], P[3]);
      o[1] = fmaf(dp[pix], d[1], P[7]);
      o[2] = fmaf(dp[pix], d[2], P[11]);
    }
  }
  free(keys);
}

/* Per-call robot-point subset (csrc/franka.hip mpx_draw_subset).  The reference redraws the column subset of the robot
 * cloud on every sampler(q) call -- robofin's FrankaSampler.sample: np.random.choice(P, num_points, replace=False), ONE
 * draw shared by the whole batch (call sites mpinets/model.py:170-181, run_inference.py:188-189).  NumPy's global RNG
 * is not reproducible on a device, so only the DISTRIBUTION is defined: n_out of `total` table rows, uniformly without
 * replacement, in uniform order.  Engine definition (restated here): row i gets the Philox4x32-10 key
 * philox(ctr = (i >> 2, draw, 11, 0), key = seed)[i & 3]; the n_out smallest (key, row) pairs win, in that order.     */
ORC_API void orc_draw_subset(int total, int n_out, uint32_t k0, uint32_t k1, uint32_t draw, int32_t *out) {
  uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)total);
  for (int i = 0; i < total; ++i) {
    uint32_t r[4];
    orc_philox((uint32_t)(i >> 2), draw, 11u, 0u, k0, k1, r);
    keys[i] = ((uint64_t)r[i & 3] << 32) | (uint32_t)i;
  }
  qsort(keys, (size_t)total, sizeof(uint64_t), orc_cmp_u64);
  for (int i = 0; i < n_out && i < total; ++i) out[i] = (int32_t)(uint32_t)keys[i];
  free(keys);
}
