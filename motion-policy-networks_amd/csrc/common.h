// common.h -- shared host/device helpers for libmpinets_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mpinets_hip.h"

#define MPX_EXPORT extern "C" __attribute__((visibility("default")))

// ---- error plumbing -------------------------------------------------------------------------
void mpx_set_error(const char *fmt, ...);

#define MPX_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      mpx_set_error(__VA_ARGS__);     \
      return 1;                       \
    }                                 \
  } while (0)

// Raises a kernel's dynamic-LDS limit to `bytes` (its largest supported launch) ONCE per (kernel, device): the first
// launch on a device pays the hipFuncSetAttribute, later calls only enqueue.  One static mask per expansion site, so
// use it once per kernel instantiation.
#define MPX_LDS_LIMIT_ONCE(kernel, bytes, what)                                                                \
  do {                                                                                                         \
    static std::atomic<unsigned long long> lds_done_{0};                                                       \
    int dev_ = 0;                                                                                              \
    (void)hipGetDevice(&dev_);                                                                                 \
    const unsigned long long bit_ = 1ull << (dev_ & 63);                                                       \
    if (!(lds_done_.load(std::memory_order_acquire) & bit_)) {                                                 \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),                              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));           \
      MPX_REQUIRE(e_ == hipSuccess, "%s: cannot reserve %d B of LDS: %s", what, (int)(bytes), hipGetErrorString(e_)); \
      lds_done_.fetch_or(bit_, std::memory_order_release);                                                     \
    }                                                                                                          \
  } while (0)

#define MPX_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      mpx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return 2;                                                             \
    }                                                                       \
    return 0;                                                               \
  } while (0)

static inline hipStream_t mpx_s(mpx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
// The 8 unit-queue counters (one per XCD) of `stream` on the current device, zeroed on that stream in front of the launch
// that uses them (sa_mlp_bf16.hip; nullptr on failure).  Persistent grouped-MLP kernels take their work units from it.
unsigned int *mpx_unit_queue_for(hipStream_t stream, int *exhausted);
int mpx_unit_queue_slots();
void mpx_unit_queue_set_slots(int n);
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- launch slabs ----------------------------------------------------------------------------------------------------
// gridDim.y is limited to 65535, and kernels that reach an operand through a 32-bit buffer offset need it under 4 GB.
// A batched entry point whose batch exceeds one launch walks it in slabs (same kernels, same per-row arithmetic: results
// do not depend on the slabbing) -- the whole of BASELINE configs[4] (65 536 environments) runs on one 288 GB GPU.
constexpr int MPX_GRID_Y = 65535;
// rows one launch of a row-blocked GEMM may cover: <= 65528 blocks of `bm` rows (a multiple of 8 blocks: the XCD-aware
// tile order keeps its shape) and rows * row_bytes < 4 GB when row_bytes > 0
static inline int64_t mpx_row_slab(int bm, int64_t row_bytes) {
  int64_t blocks = 65528;
  if (row_bytes > 0) {
    const int64_t fit = ((((int64_t)1 << 32) - 8192) / row_bytes / bm) & ~(int64_t)7;
    if (fit < blocks) blocks = fit;
  }
  if (blocks < 8) blocks = 8;
  return blocks * bm;
}

// ---- device math with a pinned evaluation order ----------------------------------------------
// Build uses -ffp-contract=off: fused multiply-adds exist only where __builtin_fmaf is written,
// so these helpers evaluate exactly like their restatement in oracle/mpn_oracle.c.

__device__ __forceinline__ float mpx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// THE squared distance of the index path (FPS, ball query; restated as sqdist() in oracle/mpn_oracle.c -- one definition
// on each side).  pointnet2_ops writes `(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)` (sampling_gpu.cu,
// ball_query_gpu.cu) and nvcc contracts it (-fmad=true).  LLVM's DAG combiner -- NVVM is built on it -- fuses the LEFT
// multiply of each add first: (a*a + b*b) -> fma(a,a, b*b), then (. + c*c) -> fma(c,c, .), i.e. the product that is
// rounded on its own is dy^2 (profiles/r03_contraction_evidence.md: the x86 and gfx950 disassembly of exactly that
// expression under -ffp-contract=fast).  -DMPX_SQDIST_XFIRST builds the order rounds 1-2 assumed (dx^2 rounded on
// its own) for A/B runs; the two orders pick a different FPS sequence on ~0.2 % of 6272-point clouds
// (tests/test_oracle_pointnet.py::test_contraction_order_ab).  Both forms are monotone in |dx|, |dy|, |dz| (the culled FPS
// kernel's box bound relies on that).
__device__ __forceinline__ float mpx_sqdist(float dx, float dy, float dz) {
#ifdef MPX_SQDIST_XFIRST
  return mpx_fma(dz, dz, mpx_fma(dy, dy, dx * dx));
#else
  return mpx_fma(dz, dz, mpx_fma(dx, dx, dy * dy));
#endif
}

// ---- segmented max-pool in a GEMM epilogue (training: the last layer of a grouped MLP + its max-pool, row N1) ----------
// Rows belong to segments (seg[row], non-decreasing); per (segment, column) the largest activated value and the FIRST row
// attaining it are kept as one 64-bit key {order-preserving bits of the value, ~row} by atomicMax -- the [rows, columns]
// matrix never reaches memory.  keys are zeroed by the launcher (0 is below every real key) and unpacked afterwards.
__device__ __forceinline__ unsigned mpx_ordered_bits(float v) {  // monotone map float -> unsigned
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float mpx_ordered_float(unsigned u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// the 128 x 128 tile kernels' accumulator layout: a lane holds column col_base + 32 j + l31 of rows
// row_base + 32 i + (r & 3) + 8 (r >> 2) + 4 half, ascending in (i, r).  val(i, j, r) = the activated value.
template <class F>
__device__ __forceinline__ void mpx_segpool_tile(F &&val, int row_base, int col_base, int M, int N, int half, int l31,
                                                 const int32_t *__restrict__ seg, int row0,
                                                 unsigned long long *__restrict__ keys, int ldk) {
  int sg[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      sg[i][r] = row < M ? seg[row] : -1;
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col_base + j * 32 + l31;
    if (col >= N) continue;
    int cs = -1;
    unsigned long long best = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int s = sg[i][r];
        if (s < 0) continue;
        const unsigned row = (unsigned)(row0 + row_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
        const unsigned long long key = ((unsigned long long)mpx_ordered_bits(val(i, j, r)) << 32) | (0xFFFFFFFFu - row);
        if (s != cs) {
          if (cs >= 0) atomicMax(keys + (size_t)cs * ldk + col, best);
          cs = s, best = key;
        } else {
          best = key > best ? key : best;
        }
      }
    if (cs >= 0) atomicMax(keys + (size_t)cs * ldk + col, best);
  }
}

// Cody-Waite by pi/2 + fixed polynomials; |x| up to ~1e3 rad is far more than joint angles need.
__device__ __forceinline__ void mpx_sincos(float x, float &s, float &c) {
  const float TWO_OVER_PI = 0.63661977236758134308f;
  const float PIO2_HI = 1.57079625129699707031f;
  const float PIO2_LO = 7.54978941586159635335e-08f;
  float kf = __builtin_rintf(x * TWO_OVER_PI);
  float r = mpx_fma(-kf, PIO2_HI, x);
  r = mpx_fma(-kf, PIO2_LO, r);
  float r2 = r * r;
  float ps = mpx_fma(r2, 2.7557314297e-06f, -1.9841270114e-04f);
  ps = mpx_fma(ps, r2, 8.3333337680e-03f);
  ps = mpx_fma(ps, r2, -1.6666667163e-01f);
  float sn = mpx_fma(r * r2, ps, r);
  float pc = mpx_fma(r2, -2.7557314297e-07f, 2.4801587642e-05f);
  pc = mpx_fma(pc, r2, -1.3888889225e-03f);
  pc = mpx_fma(pc, r2, 4.1666667908e-02f);
  pc = mpx_fma(pc, r2, -0.5f);
  float cs = mpx_fma(pc, r2, 1.0f);
  int k = ((int)kf) & 3;
  s = (k & 1) ? cs : sn;
  c = (k & 1) ? sn : cs;
  if (k == 1 || k == 2) c = -c;
  if (k >= 2) s = -s;
}

// max of a value with its partner lane 32 apart (lane i <-> i ^ 32), in every lane: one v_permlane32_swap (VALU)
// instead of a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ float mpx_max_across_halves(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r[0] = {lo, lo}, r[1] = {hi, hi}
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// 3x4 rigid transform: r[9] row-major rotation, t[3]
struct Rigid {
  float r[9];
  float t[3];
};

__device__ __forceinline__ Rigid rigid_compose(const Rigid &a, const float (&fr)[9], const float (&ft)[3]) {
  Rigid o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = a.r[3 * i + 0] * fr[0 + c];
      acc = mpx_fma(a.r[3 * i + 1], fr[3 + c], acc);
      acc = mpx_fma(a.r[3 * i + 2], fr[6 + c], acc);
      o.r[3 * i + c] = acc;
    }
    float acc = a.t[i];
    acc = mpx_fma(a.r[3 * i + 0], ft[0], acc);
    acc = mpx_fma(a.r[3 * i + 1], ft[1], acc);
    acc = mpx_fma(a.r[3 * i + 2], ft[2], acc);
    o.t[i] = acc;
  }
  return o;
}

__device__ __forceinline__ Rigid rigid_rotz(const Rigid &p, float s, float c) {
  Rigid o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float a = p.r[3 * i + 0], b = p.r[3 * i + 1];
    o.r[3 * i + 0] = mpx_fma(b, s, a * c);
    o.r[3 * i + 1] = mpx_fma(b, c, -(a * s));
    o.r[3 * i + 2] = p.r[3 * i + 2];
    o.t[i] = p.t[i];
  }
  return o;
}

__device__ __forceinline__ void rigid_apply(const float *f /*12*/, float x, float y, float z, float &ox,
                                            float &oy, float &oz) {
  float a0 = f[0] * x;
  a0 = mpx_fma(f[1], y, a0);
  a0 = mpx_fma(f[2], z, a0);
  float a1 = f[3] * x;
  a1 = mpx_fma(f[4], y, a1);
  a1 = mpx_fma(f[5], z, a1);
  float a2 = f[6] * x;
  a2 = mpx_fma(f[7], y, a2);
  a2 = mpx_fma(f[8], z, a2);
  ox = a0 + f[9];
  oy = a1 + f[10];
  oz = a2 + f[11];
}

// Franka Panda chain (public URDF constants).  `visit(id, frame)` is called once per frame, in chain order
// (0 = link0 ... 8 = link8, 9 hand, 10 / 11 fingers, 12 / 13 fingertips, 14 right_gripper), with the frame in registers:
// a caller that consumes a frame as soon as it exists (the lanes-as-waypoints collision kernel) never stores the chain.
template <class Visit>
__device__ __forceinline__ void franka_fk_visit(const float *q7, float finger, Visit &&visit) {
  constexpr float SH = 0.70710678118654752440f;
  Rigid cur = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}};
  visit(std::integral_constant<int, 0>{}, cur);
  const float JR[7][9] = {
      {1, 0, 0, 0, 1, 0, 0, 0, 1},  {1, 0, 0, 0, 0, 1, 0, -1, 0}, {1, 0, 0, 0, 0, -1, 0, 1, 0},
      {1, 0, 0, 0, 0, -1, 0, 1, 0}, {1, 0, 0, 0, 0, 1, 0, -1, 0}, {1, 0, 0, 0, 0, -1, 0, 1, 0},
      {1, 0, 0, 0, 0, -1, 0, 1, 0},
  };
  const float JT[7][3] = {
      {0.0f, 0.0f, 0.333f},     {0.0f, 0.0f, 0.0f}, {0.0f, -0.316f, 0.0f}, {0.0825f, 0.0f, 0.0f},
      {-0.0825f, 0.384f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.088f, 0.0f, 0.0f},
  };
  auto joint = [&](auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    float s, c;
    mpx_sincos(q7[j], s, c);
    Rigid tmp = rigid_compose(cur, JR[j], JT[j]);
    cur = rigid_rotz(tmp, s, c);
    visit(std::integral_constant<int, j + 1>{}, cur);
  };
  joint(std::integral_constant<int, 0>{});
  joint(std::integral_constant<int, 1>{});
  joint(std::integral_constant<int, 2>{});
  joint(std::integral_constant<int, 3>{});
  joint(std::integral_constant<int, 4>{});
  joint(std::integral_constant<int, 5>{});
  joint(std::integral_constant<int, 6>{});
  const float I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const float T8[3] = {0.0f, 0.0f, 0.107f};
  Rigid l8 = rigid_compose(cur, I3, T8);
  visit(std::integral_constant<int, 8>{}, l8);
  const float RH[9] = {SH, SH, 0, -SH, SH, 0, 0, 0, 1};
  const float Z3[3] = {0.0f, 0.0f, 0.0f};
  Rigid hand = rigid_compose(l8, RH, Z3);
  visit(std::integral_constant<int, 9>{}, hand);
  const float TL[3] = {0.0f, finger, 0.0584f};
  const float TR[3] = {0.0f, -finger, 0.0584f};
  Rigid lf = rigid_compose(hand, I3, TL);
  Rigid rf = rigid_compose(hand, I3, TR);
  visit(std::integral_constant<int, 10>{}, lf);
  visit(std::integral_constant<int, 11>{}, rf);
  const float TT[3] = {0.0f, 0.0f, 0.045f};
  visit(std::integral_constant<int, 12>{}, rigid_compose(lf, I3, TT));
  visit(std::integral_constant<int, 13>{}, rigid_compose(rf, I3, TT));
  const float RG[9] = {-SH, -SH, 0, SH, -SH, 0, 0, 0, 1};
  const float TG[3] = {0.0f, 0.0f, 0.1f};
  visit(std::integral_constant<int, 14>{}, rigid_compose(l8, RG, TG));
}

// Writes the 15 frames x 12 floats to `out` (any address space the caller indexes with a plain pointer: LDS or global).
__device__ __forceinline__ void franka_fk_frames(const float *q7, float finger, float *out) {
  franka_fk_visit(q7, finger, [&](auto ID, const Rigid &g) __attribute__((always_inline)) {
    constexpr int id = decltype(ID)::value;
#pragma unroll
    for (int k = 0; k < 9; ++k) out[12 * id + k] = g.r[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[12 * id + 9 + k] = g.t[k];
  });
}
