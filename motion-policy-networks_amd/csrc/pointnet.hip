// pointnet.hip -- farthest-point sampling, ball query and neighbourhood grouping.
//
// Replaces the pointnet2_ops v3.2.0 CUDA extension the reference pip-installs
// (/root/reference/docker/Dockerfile:152; call sites mpinets/model.py:27,366-383,423-424).
// Index semantics are those of that extension (restated in oracle/mpn_oracle.c):
//   FPS   start at index 0; running distance 1e10; points with |p|^2 <= 1e-3 never update and
//         are never candidates; the arg-max over d2 breaks ties the way the reference's strided
//         thread scan (thread t = k mod bs keeps its first maximum, bs = opt_n_threads(N)) followed
//         by its shared-memory tree reduction (slot t absorbs slot t+s, ties keep slot t,
//         s = bs/2..1) does: the smallest BIT-REVERSED thread id wins, then the smallest k;
//   ball  first `nsample` indices in ascending order with d2 < r^2, all slots pre-filled with
//         the first hit, zeros when nothing is in range.
// Distances are mpx_sqdist() (common.h): fma(dz,dz,fma(dx,dx,dy*dy)), upstream's expression as LLVM / NVVM contracts it.
//
// CDNA4 design: one workgroup per environment; the whole cloud lives in registers
// (x,y,z,running distance: PTS points per lane) with an LDS copy used only to broadcast the
// coordinates of the point just selected.  Each of the npoint-1 dependent iterations is one
// pass of VALU work, a DPP (cross-lane) arg-max inside every wave and ONE barrier: waves publish
// a 64-bit key {float bits of d2, ~tie-rank} into a double-buffered LDS slot array and every wave
// re-reduces the <=16 slots itself.  The kernel is latency/LDS-bound, not HBM-bound: HBM traffic
// is the 12 (or 16) bytes per point read once.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

typedef unsigned long long u64;

// ---- DPP helpers --------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ u64 dpp_mov64(u64 v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);
  hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
  return ((u64)hi << 32) | lo;
}
// Keys are {float bits of d2 (non-negative, finite), 32-bit tie rank}: as IEEE doubles such patterns
// are positive and finite (or subnormal -- f64 denormals are preserved in the kernel mode), and
// positive doubles order like their bit patterns, so one v_max_f64 replaces the three-instruction
// 64-bit integer compare-and-select.
struct U32Pair {
  unsigned lo, hi;
};
__device__ __forceinline__ u64 pack64(unsigned lo, unsigned hi) { return __builtin_bit_cast(u64, U32Pair{lo, hi}); }
__device__ __forceinline__ u64 umax64(u64 a, u64 b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, a)), "v"(__builtin_bit_cast(double, b)));
  return __builtin_bit_cast(u64, r);
}

// max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ u64 row_max64(u64 v) {
  v = umax64(v, dpp_mov64<0xB1>(v));   // quad_perm [1,0,3,2]
  v = umax64(v, dpp_mov64<0x4E>(v));   // quad_perm [2,3,0,1]
  v = umax64(v, dpp_mov64<0x141>(v));  // row_half_mirror
  v = umax64(v, dpp_mov64<0x140>(v));  // row_mirror
  return v;
}
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_max64(u64 v) {
  v = row_max64(v);
  u64 a = umax64(readlane64(v, 0), readlane64(v, 16));
  u64 b = umax64(readlane64(v, 32), readlane64(v, 48));
  return umax64(a, b);
}

// ---- farthest point sampling --------------------------------------------------------------------
// grid B, block = min(512, roundup64(N)) (MPX_FPS_BLOCK overrides, tuning only); dynamic LDS = 3*N floats + slots.
// Two workgroups must share a CU (their serial pick chains interleave): 1024-thread blocks need
// <= 64 VGPRs (8 waves/SIMD), 512-thread blocks with up to 16 points per lane <= 128.
template <int PTS>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PTS <= 8 ? 8 : 4, PTS <= 8 ? 8 : 4)))
    fps_kernel(const float *__restrict__ xyz, int N, int stride, int npoint, int log2bs,
               int32_t *__restrict__ idx, float *__restrict__ new_xyz, int new_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *slots = reinterpret_cast<u64 *>(smem);              // [2][16]
  float *sx = reinterpret_cast<float *>(smem + 256);       // [N]
  float *sy = sx + N;
  float *sz = sy + N;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float *pts = xyz + (size_t)b * N * stride;
  int32_t *out = idx + (size_t)b * npoint;
  float *nxyz = new_xyz ? new_xyz + (size_t)b * npoint * new_stride : nullptr;
  const unsigned bsmask = (1u << log2bs) - 1u;

  // per point one 64-bit key {low: tie rank (constant), high: bits of its running min distance} kept in one
  // register pair: the v_min overwrites the high half in place and the pair feeds v_max_f64 directly
  float x[PTS], y[PTS], z[PTS];
  u64 key[PTS];
#pragma unroll
  for (int i = 0; i < PTS; ++i) {
    const int k = tid + i * nthr;
    x[i] = y[i] = z[i] = 0.0f;
    key[i] = 0;  // a point that may never be picked keeps key == 0 (distance +0, rank 0)
    if (k < N) {
      x[i] = pts[(size_t)k * stride + 0];
      y[i] = pts[(size_t)k * stride + 1];
      z[i] = pts[(size_t)k * stride + 2];
      sx[k] = x[i];
      sy[k] = y[i];
      sz[k] = z[i];
      const float mag = mpx_sqdist(x[i], y[i], z[i]);
      // upstream compares the float magnitude with the DOUBLE literal 1e-3 (`if (mag <= 1e-3) continue;`): a point with
      // mag == 1e-3f (= 0.0010000000475 > 1e-3) is kept
      if (!((double)mag <= 1e-3)) {
        // tie order of the reference: smaller (bitrev(k mod bs), k / bs) wins -> larger key wins.
        // brev of a < 2^9 value lands in the top 9 bits: keeping the top 16 preserves the order.
        const unsigned rank = (__brev((unsigned)k & bsmask) & 0xFFFF0000u) | ((unsigned)k >> log2bs);
        key[i] = pack64(0xFFFFFFFFu - rank, __float_as_uint(1e10f));
      }
    }
  }
  if (tid < 32) slots[tid] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < npoint; ++j) {
    const float x1 = sx[old], y1 = sy[old], z1 = sz[old];
    if (tid == 0) {
      out[j - 1] = old;
      if (nxyz) {
        nxyz[(size_t)(j - 1) * new_stride + 0] = x1;
        nxyz[(size_t)(j - 1) * new_stride + 1] = y1;
        nxyz[(size_t)(j - 1) * new_stride + 2] = z1;
      }
    }
    u64 best = 0;
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
      const float dx = x[i] - x1, dy = y[i] - y1, dz = z[i] - z1;
      const float d = mpx_sqdist(dx, dy, dz);
      float d2;  // min(d, temp[k]) as one v_min_f32 (fminf() adds a canonicalising v_max per call)
      asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(__uint_as_float((unsigned)(key[i] >> 32))));
      // a point that is not a candidate has rank 0 and distance +0, so its whole key is 0
      key[i] = pack64((unsigned)key[i], __float_as_uint(d2));
      best = umax64(best, key[i]);
    }
    best = wave_max64(best);
    u64 *slot = slots + (j & 1) * 16;
    if (lane == 0) slot[wave] = best;
    __syncthreads();
    u64 m = row_max64(slot[lane & 15]);
    m = readlane64(m, 0);
    if (m == 0) {
      old = 0;  // nothing was a candidate: the reference's besti stays 0
    } else {
      const unsigned rank = 0xFFFFFFFFu - (unsigned)m;
      old = (int)(((rank & 0xFFFFu) << log2bs) | __brev(rank & 0xFFFF0000u));
    }
  }
  if (tid == 0 && npoint > 0) {
    out[npoint - 1] = old;
    if (nxyz) {
      nxyz[(size_t)(npoint - 1) * new_stride + 0] = sx[old];
      nxyz[(size_t)(npoint - 1) * new_stride + 1] = sy[old];
      nxyz[(size_t)(npoint - 1) * new_stride + 2] = sz[old];
    }
  }
}

// ---- farthest point sampling of a SMALL cloud (N <= 512: the second set-abstraction module samples 128 of 512) -------
// One WAVE per environment, the whole cloud in its registers (8 points per lane): a pick is one pass of VALU work, one
// DPP arg-max inside the wave and three v_readlane's for the picked point's coordinates -- no LDS, no barrier, and up to
// 32 environments per CU instead of 4.  Same arithmetic, same 64-bit keys and tie ranks as fps_kernel.
template <int PTS>
__global__ void __launch_bounds__(64)
    fps_wave_kernel(const float *__restrict__ xyz, int N, int stride, int npoint, int log2bs, int32_t *__restrict__ idx,
                    float *__restrict__ new_xyz, int new_stride) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float *pts = xyz + (size_t)b * N * stride;
  int32_t *out = idx + (size_t)b * npoint;
  float *nxyz = new_xyz ? new_xyz + (size_t)b * npoint * new_stride : nullptr;
  const unsigned bsmask = (1u << log2bs) - 1u;
  float x[PTS], y[PTS], z[PTS];
  u64 key[PTS];
#pragma unroll
  for (int i = 0; i < PTS; ++i) {
    const int k = lane + i * 64;
    x[i] = y[i] = z[i] = 0.0f;
    key[i] = 0;
    if (k < N) {
      x[i] = pts[(size_t)k * stride + 0];
      y[i] = pts[(size_t)k * stride + 1];
      z[i] = pts[(size_t)k * stride + 2];
      const float mag = mpx_sqdist(x[i], y[i], z[i]);
      if (!((double)mag <= 1e-3)) {
        const unsigned rank = (__brev((unsigned)k & bsmask) & 0xFFFF0000u) | ((unsigned)k >> log2bs);
        key[i] = pack64(0xFFFFFFFFu - rank, __float_as_uint(1e10f));
      }
    }
  }
  // coordinates of point `old` (wave-uniform): slot old / 64 of lane old % 64
  auto coords = [&](int old, float &ox, float &oy, float &oz) __attribute__((always_inline)) {
    const int slot = old >> 6, ln = old & 63;
    float sx = x[0], sy = y[0], sz = z[0];
#pragma unroll
    for (int i = 1; i < PTS; ++i) {
      sx = slot == i ? x[i] : sx;
      sy = slot == i ? y[i] : sy;
      sz = slot == i ? z[i] : sz;
    }
    ox = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(sx), ln));
    oy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(sy), ln));
    oz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(sz), ln));
  };
  int old = 0;
  for (int j = 1; j < npoint; ++j) {
    float x1, y1, z1;
    coords(old, x1, y1, z1);
    if (lane == 0) {
      out[j - 1] = old;
      if (nxyz) {
        nxyz[(size_t)(j - 1) * new_stride + 0] = x1;
        nxyz[(size_t)(j - 1) * new_stride + 1] = y1;
        nxyz[(size_t)(j - 1) * new_stride + 2] = z1;
      }
    }
    u64 best = 0;
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
      const float dx = x[i] - x1, dy = y[i] - y1, dz = z[i] - z1;
      const float d = mpx_sqdist(dx, dy, dz);
      float d2;
      asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(__uint_as_float((unsigned)(key[i] >> 32))));
      key[i] = pack64((unsigned)key[i], __float_as_uint(d2));
      best = umax64(best, key[i]);
    }
    const u64 m = wave_max64(best);
    if (m == 0) {
      old = 0;
    } else {
      const unsigned rank = 0xFFFFFFFFu - (unsigned)m;
      old = (int)(((rank & 0xFFFFu) << log2bs) | __brev(rank & 0xFFFF0000u));
    }
  }
  if (npoint > 0) {
    float x1, y1, z1;
    coords(old, x1, y1, z1);
    if (lane == 0) {
      out[npoint - 1] = old;
      if (nxyz) {
        nxyz[(size_t)(npoint - 1) * new_stride + 0] = x1;
        nxyz[(size_t)(npoint - 1) * new_stride + 1] = y1;
        nxyz[(size_t)(npoint - 1) * new_stride + 2] = z1;
      }
    }
  }
}

// ---- farthest point sampling with exact culling (512 < N <= 8192) ---------------------------------------------------
// fps_kernel above is bound by its VALU stream: every pick re-evaluates all N running distances (7 instructions per
// point) although, after the first few dozen picks, a new pick lowers the running distance of only the points near it.
// Here the cloud is first ordered along a Morton curve (16^3 cells over its bounding box, counting sort in LDS) and cut
// into chunks of 64 consecutive points -- one register slot of one wave, dealt round-robin to the 8 waves so that the
// chunks around a pick belong to different waves.  Per chunk a wave keeps, in lane `slot` of a few registers, the
// bounding box and an UPPER bound `cub` of the chunk's largest running distance.  A pick costs a wave one VALU pass
// over those <= 16 lanes: the smallest and the largest squared distance from the pick to the box; the chunk's points
// are re-evaluated only if the smallest is below `cub` (28 of 98 chunks on the bench scenes), and `cub` drops to the
// largest (every running distance of the chunk is now <= its distance to the pick <= that).  Skipping is exact: the
// lower bound is evaluated with the same fp32 operations in the same order as a point's distance (x - x1 is monotone
// in x; a*a and fma(a,a,c) are monotone in |a| and c), so bound <= distance(p) for every p in the box IN fp32, and
// bound >= cub >= running(p) means min(distance, running) == running.  (`cub` is padded by 1e-6 relative: it only has to
// be an upper bound.)  The arg-max over all keys, the tie order and the outputs are those of fps_kernel bit for bit:
// the key's low word carries the same tie rank (bit-reversed k mod 512, then k / 512) in its upper bits; the point's
// position in the sorted order rides in the 13 bits below, which can never decide a comparison (the rank bits above
// identify the point) and gives the picked point's coordinates with one LDS lookup.
// A first form kept each chunk's exact (distance, rank) maximum by a wave reduction per re-evaluated chunk (9.7 chunks
// per pick instead of 28): bit-exact too, and no faster than fps_kernel -- the serial reductions made the pick a
// 1.7 k-cycle latency chain, two of which fit a CU.
template <int CTRL>
__device__ __forceinline__ float dpp_movf(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
template <bool MAX>
__device__ __forceinline__ float wave_red_f32(float v) {  // min / max over the wave, in every lane's return value
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
  v = op(v, dpp_movf<0xB1>(v));
  v = op(v, dpp_movf<0x4E>(v));
  v = op(v, dpp_movf<0x141>(v));
  v = op(v, dpp_movf<0x140>(v));
  auto rl = [&](int l) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); };
  return op(op(rl(0), rl(16)), op(rl(32), rl(48)));
}
// v with lane `lane` replaced by the wave-uniform value s.  (v_writelane takes one SGPR: the lane select must be an
// inline constant -- the switch folds away where `lane` is an unrolled loop index.)
__device__ __forceinline__ unsigned writelane_u32(unsigned s, int lane, unsigned v) {
  const unsigned su = (unsigned)__builtin_amdgcn_readfirstlane((int)s);
#define MPX_WL(L) case L: asm("v_writelane_b32 %0, %1, " #L : "+v"(v) : "s"(su)); break;
  switch (lane) {
    MPX_WL(0) MPX_WL(1) MPX_WL(2) MPX_WL(3) MPX_WL(4) MPX_WL(5) MPX_WL(6) MPX_WL(7)
    MPX_WL(8) MPX_WL(9) MPX_WL(10) MPX_WL(11) MPX_WL(12) MPX_WL(13) MPX_WL(14) MPX_WL(15)
    MPX_WL(16) MPX_WL(17) MPX_WL(18) MPX_WL(19) MPX_WL(20) MPX_WL(21) MPX_WL(22) MPX_WL(23)
    MPX_WL(24) MPX_WL(25) MPX_WL(26) MPX_WL(27) MPX_WL(28) MPX_WL(29) MPX_WL(30) MPX_WL(31)
    default: break;
  }
#undef MPX_WL
  return v;
}
__device__ __forceinline__ float writelane_f32(float s, int lane, float v) {
  return __uint_as_float(writelane_u32(__float_as_uint(s), lane, __float_as_uint(v)));
}
__device__ __forceinline__ u64 writelane64(u64 s, int lane, u64 v) {
  return pack64(writelane_u32((unsigned)s, lane, (unsigned)v), writelane_u32((unsigned)(s >> 32), lane, (unsigned)(v >> 32)));
}
__device__ __forceinline__ unsigned morton4(unsigned v) {  // 4 bits -> bits 0, 3, 6, 9
  v &= 0xFu;
  v = (v | (v << 4)) & 0xC3u;
  v = (v | (v << 2)) & 0x249u;
  return v;
}

// loop with compile-time indices (arrays indexed this way are split into registers before any control flow is built
// around their elements; with `#pragma unroll` loops the 26-dword key array of fps_cull_kernel stayed ONE vector value
// that was copied and spilled whole at every conditional update)
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}

// max of an unsigned over the wave as six in-place DPP-fused v_max_u32 (row steps, then row_bcast:15 / :31 carry the row
// results upwards: lane 63 holds the maximum) + one v_readlane -- against 12 DPP moves, 7 v_max_f64, 8 v_readlane and 8
// moves for the 64-bit key maximum of wave_max64.  (Inline asm: the two wait states a DPP read needs after the VALU
// write of its source are written out.)
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v) {
  asm volatile(
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// arg-max of 64-bit keys {low: tie word, high: bits of a non-negative float} over the wave, in two 32-bit stages: the
// largest high word, then the largest low word among the lanes that hold it (the same total order as the 64-bit max)
__device__ __forceinline__ u64 wave_max64_staged(u64 k) {
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned mhi = wave_max_u32_dpp(hi);
  const unsigned mlo = wave_max_u32_dpp(hi == mhi ? lo : 0u);
  return pack64(mlo, mhi);
}

constexpr int FPSC_CELLS = 4096;
// dynamic LDS: [0,256) reduction slots, [256,768) scalars (position of point 0, cloud bounding box exchange), then the
// cloud in sorted order sx | sy | sz (3 N floats); the prologue's histogram (16 KB) and position -> index map (2 N bytes)
// live in the same area before the coordinates are written
__host__ __device__ constexpr size_t fpsc_lds_bytes(int N, bool nocloud = false) {
  const size_t cloud = nocloud ? 0 : (size_t)3 * N * 4, pro = (size_t)FPSC_CELLS * 4 + (((size_t)N * 2 + 15) & ~(size_t)15);
  return 768 + (cloud > pro ? cloud : pro);
}
template <int PTS, int FPSC_WAVES>
__global__ void __launch_bounds__(64 * FPSC_WAVES)
    __attribute__((amdgpu_waves_per_eu(FPSC_WAVES == 4 ? 3 : FPSC_WAVES / 2, FPSC_WAVES == 4 ? 3 : FPSC_WAVES / 2)))
    fps_cull_kernel(const float *__restrict__ xyz, int N, int stride, int npoint, int32_t *__restrict__ idx,
                    float *__restrict__ new_xyz, int new_stride) {
  static_assert(PTS >= 2 && PTS <= 32, "slots per lane");
  static_assert(FPSC_WAVES == 4 || FPSC_WAVES == 8 || FPSC_WAVES == 16, "4 x 25, 8 x 13 or 16 x 7 points per lane");
  constexpr int FPSC_THREADS = 64 * FPSC_WAVES, CPT = FPSC_CELLS / FPSC_THREADS;  // cells per thread in the scan
  // NOCLOUD (the 4-wave form): no sorted copy of the cloud in LDS; the coordinates of a pick are fetched from the cloud
  // in memory by a wave-uniform (scalar) load.  LDS = the prologue's sort scratch (29 KB) instead of 76 KB: three
  // workgroups per CU.
  constexpr bool NOCLOUD = FPSC_WAVES == 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *slots = reinterpret_cast<u64 *>(smem);                    // [2][16]
  int *scal = reinterpret_cast<int *>(smem + 256);               // [0]: sorted position of point 0
  float *bbx = reinterpret_cast<float *>(smem + 256 + 16);       // [waves][6] (16 waves: ends at byte 656)
  unsigned *hist = reinterpret_cast<unsigned *>(smem + 768);     // [4096]   (prologue)
  unsigned short *pmap = reinterpret_cast<unsigned short *>(smem + 768 + FPSC_CELLS * 4);  // [N] (prologue)
  float *sx = reinterpret_cast<float *>(smem + 768), *sy = sx + N, *sz = sy + N;           // (after the prologue)

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pts = xyz + (size_t)b * N * stride;
  int32_t *out = idx + (size_t)b * npoint;
  float *nxyz = new_xyz ? new_xyz + (size_t)b * npoint * new_stride : nullptr;
  const float INF = __builtin_inff();

  // ---- prologue 1: cells of the points in index order (thread t owns k = t + 512 i), histogram, positions ----------
  unsigned code[PTS], rnk[PTS];
  {
    float x[PTS], y[PTS], z[PTS];
    float mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
      const int k = tid + i * FPSC_THREADS;
      x[i] = y[i] = z[i] = 0.0f;
      if (k < N) {
        x[i] = pts[(size_t)k * stride + 0];
        y[i] = pts[(size_t)k * stride + 1];
        z[i] = pts[(size_t)k * stride + 2];
        mn[0] = fminf(mn[0], x[i]), mn[1] = fminf(mn[1], y[i]), mn[2] = fminf(mn[2], z[i]);
        mx[0] = fmaxf(mx[0], x[i]), mx[1] = fmaxf(mx[1], y[i]), mx[2] = fmaxf(mx[2], z[i]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = wave_red_f32<false>(mn[c]);
      mx[c] = wave_red_f32<true>(mx[c]);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) bbx[wave * 6 + c] = mn[c], bbx[wave * 6 + 3 + c] = mx[c];
    }
    for (int i = tid; i < FPSC_CELLS; i += FPSC_THREADS) hist[i] = 0;
    if (tid < 32) slots[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < FPSC_WAVES; ++w)
#pragma unroll
      for (int c = 0; c < 3; ++c) mn[c] = fminf(mn[c], bbx[w * 6 + c]), mx[c] = fmaxf(mx[c], bbx[w * 6 + 3 + c]);
    float sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) sc[c] = mx[c] > mn[c] ? 16.0f * __builtin_amdgcn_rcpf(mx[c] - mn[c]) : 0.0f;  // (cell choice only affects speed)
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
      const int k = tid + i * FPSC_THREADS;
      code[i] = rnk[i] = 0;
      if (k < N) {
        const int cx = min(15, max(0, (int)((x[i] - mn[0]) * sc[0]))), cy = min(15, max(0, (int)((y[i] - mn[1]) * sc[1]))),
                  cz = min(15, max(0, (int)((z[i] - mn[2]) * sc[2])));
        code[i] = morton4(cx) | (morton4(cy) << 1) | (morton4(cz) << 2);
        rnk[i] = atomicAdd(&hist[code[i]], 1u);
      }
    }
    __syncthreads();
  }
  {  // exclusive scan of the 4096 cell counts: thread t owns CPT consecutive cells
    unsigned c8[CPT], tot = 0;
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      c8[e] = hist[tid * CPT + e];
      tot += c8[e];
    }
    unsigned inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    unsigned *wtot = reinterpret_cast<unsigned *>(bbx);  // (the box exchange is over)
    __syncthreads();
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    unsigned base = inc - tot;
#pragma unroll
    for (int w = 0; w < FPSC_WAVES; ++w)
      if (w < wave) base += wtot[w];
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      hist[tid * CPT + e] = base;
      base += c8[e];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < PTS; ++i) {
    const int k = tid + i * FPSC_THREADS;
    if (k < N) {
      const unsigned pos = hist[code[i]] + rnk[i];
      pmap[pos] = (unsigned short)k;
      if (k == 0) scal[0] = (int)pos;
    }
  }
  __syncthreads();

  // ---- prologue 2: the points in sorted order; chunk c = positions [64 c, 64 c + 64) belongs to wave c % 8, slot c / 8 --
  float x[PTS], y[PTS], z[PTS];
  u64 key[PTS];
  int kk[PTS];
  static_for<0, PTS>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const int pos = (wave + FPSC_WAVES * i) * 64 + lane;
    kk[i] = pos < N ? (int)pmap[pos] : -1;
  });
  const int pos0 = scal[0];
  __syncthreads();  // (the map and the histogram are dead: their bytes become the sorted cloud)
  static_for<0, PTS>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const int pos = (wave + FPSC_WAVES * i) * 64 + lane;
    x[i] = y[i] = z[i] = 0.0f;
    key[i] = 0;  // a point that may never be picked keeps key == 0 (distance +0, rank 0)
    if (kk[i] >= 0) {
      const int k = kk[i];
      x[i] = pts[(size_t)k * stride + 0];
      y[i] = pts[(size_t)k * stride + 1];
      z[i] = pts[(size_t)k * stride + 2];
      if constexpr (!NOCLOUD) {
        sx[pos] = x[i];
        sy[pos] = y[i];
        sz[pos] = z[i];
      }
      const float mag = mpx_sqdist(x[i], y[i], z[i]);
      if (!((double)mag <= 1e-3)) {  // (see fps_kernel)
        // tie order of the reference: smaller (bitrev(k mod 512), k / 512) wins -> larger key wins
        const unsigned rank = (__brev((unsigned)k & 511u) & 0xFF800000u) | (((unsigned)k >> 9) << 18) | (unsigned)pos;
        key[i] = pack64(0xFFFFFFFFu - rank, __float_as_uint(1e10f));
      }
    }
  });
  // chunk records, slot i in lane i: bounding box of the chunk's points, upper bound of their running distances
  // (an empty or candidate-free chunk: 0, never re-evaluated)
  float bnx = INF, bny = INF, bnz = INF, bxx = -INF, bxy = -INF, bxz = -INF, cub = 0.0f;
  __syncthreads();  // sorted cloud visible
  // lane i < PTS walks ITS chunk's 64 points in LDS: no cross-lane traffic (positions past the end of the cloud are
  // clamped to its last point: a repeated point changes neither the box nor `any`; a chunk wholly past the end keeps
  // the inverted box and cub = 0)
  if constexpr (NOCLOUD) {
    // chunk i of this wave = register slot i of its 64 lanes: the box is six wave reductions per slot, parked in lane i
    static_for<0, PTS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const bool real = (wave + FPSC_WAVES * i) * 64 + lane < N;  // (lanes past the end of the cloud hold no point)
      const float mnx_ = wave_red_f32<false>(real ? x[i] : INF), mny_ = wave_red_f32<false>(real ? y[i] : INF),
                  mnz_ = wave_red_f32<false>(real ? z[i] : INF), mxx_ = wave_red_f32<true>(real ? x[i] : -INF),
                  mxy_ = wave_red_f32<true>(real ? y[i] : -INF), mxz_ = wave_red_f32<true>(real ? z[i] : -INF);
      const bool any = __builtin_amdgcn_ballot_w64(key[i] != 0) != 0;
      if (lane == i) {
        bnx = mnx_, bny = mny_, bnz = mnz_, bxx = mxx_, bxy = mxy_, bxz = mxz_;
        cub = any ? 1e10f : 0.0f;
      }
    });
  } else if (lane < PTS) {
    const int p0 = (wave + FPSC_WAVES * lane) * 64;
    if (p0 < N) {
      bool any = false;
#pragma unroll 8
      for (int q = 0; q < 64; ++q) {
        const int pos = min(p0 + ((q + 5 * lane) & 63), N - 1);  // (staggered: the lanes' chunks are 2 KB apart = one bank)
        const float px = sx[pos], py = sy[pos], pz = sz[pos];
        bnx = fminf(bnx, px), bny = fminf(bny, py), bnz = fminf(bnz, pz);
        bxx = fmaxf(bxx, px), bxy = fmaxf(bxy, py), bxz = fmaxf(bxz, pz);
        const float mag = mpx_sqdist(px, py, pz);
        any = any || !((double)mag <= 1e-3);
      }
      cub = any ? 1e10f : 0.0f;
    }
  }

  int old = 0, pos_old = pos0;
  // NOCLOUD: the coordinates of the current pick travel in wave-uniform registers (first pick: point 0 itself)
  float cx1 = pts[0], cy1 = pts[1], cz1 = pts[2];
  for (int j = 1; j < npoint; ++j) {
    float x1, y1, z1;
    if constexpr (NOCLOUD) {
      x1 = cx1, y1 = cy1, z1 = cz1;
    } else {
      x1 = sx[pos_old], y1 = sy[pos_old], z1 = sz[pos_old];
    }
    if (tid == 0) {
      out[j - 1] = old;
      if (nxyz) {
        nxyz[(size_t)(j - 1) * new_stride + 0] = x1;
        nxyz[(size_t)(j - 1) * new_stride + 1] = y1;
        nxyz[(size_t)(j - 1) * new_stride + 2] = z1;
      }
    }
    // smallest / largest squared distance from the pick to chunk `lane`'s box, in a point's own operation order
    const float ax = bnx - x1, bx_ = x1 - bxx, ay = bny - y1, by_ = y1 - bxy, az = bnz - z1, bz_ = z1 - bxz;
    const float ex = fmaxf(fmaxf(ax, bx_), 0.0f), ey = fmaxf(fmaxf(ay, by_), 0.0f), ez = fmaxf(fmaxf(az, bz_), 0.0f);
    const float bound = mpx_sqdist(ex, ey, ez);
    const unsigned tmask = (unsigned)__builtin_amdgcn_ballot_w64(lane < PTS && bound < cub);
    {  // |p - x1| <= max(x1 - bmin, bmax - x1) per axis
      const float fx = fmaxf(-ax, -bx_), fy = fmaxf(-ay, -by_), fz = fmaxf(-az, -bz_);
      const float far = mpx_sqdist(fx, fy, fz) * 1.000001f;
      cub = fminf(cub, far);  // (an empty chunk's box is inverted: far = +inf, cub stays 0)
    }
    static_for<0, PTS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (tmask & (1u << i)) {  // (wave-uniform)
        const float dx = x[i] - x1, dy = y[i] - y1, dz = z[i] - z1;
        const float d = mpx_sqdist(dx, dy, dz);
        float d2;  // min(d, temp[k]) as one v_min_f32 (fminf() adds a canonicalising v_max per call)
        asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(__uint_as_float((unsigned)(key[i] >> 32))));
        key[i] = pack64((unsigned)key[i], __float_as_uint(d2));
      }
    });
    u64 best;
    {  // this lane's largest key: two interleaved chains of v_max_f64
      u64 b0 = key[0], b1 = key[1];
      static_for<1, (PTS + 1) / 2>([&](auto I) {
        constexpr int i = 2 * decltype(I)::value;
        b0 = umax64(b0, key[i]);
        if constexpr (i + 1 < PTS) b1 = umax64(b1, key[i + 1]);
      });
      best = umax64(b0, b1);
    }
    best = wave_max64_staged(best);
    u64 *slot = slots + (j & 1) * 16;
    if (lane == 0) slot[wave] = best;
    __syncthreads();
    u64 m = row_max64(slot[lane & 15]);
    m = readlane64(m, 0);
    if (m == 0) {
      old = 0;  // nothing was a candidate: the reference's besti stays 0
      pos_old = pos0;
    } else {
      const unsigned rank = 0xFFFFFFFFu - (unsigned)m;
      old = (int)((((rank >> 18) & 0x1Fu) << 9) | __brev(rank & 0xFF800000u));
      pos_old = (int)(rank & 0x1FFFu);
    }
    if constexpr (NOCLOUD) {  // the pick's coordinates: a wave-uniform (scalar) load of its row -- the cloud is L2-hot
      const float *pp = pts + (size_t)old * stride;
      cx1 = pp[0], cy1 = pp[1], cz1 = pp[2];
    }
  }
  if (tid == 0 && npoint > 0) {
    out[npoint - 1] = old;
    if (nxyz) {
      nxyz[(size_t)(npoint - 1) * new_stride + 0] = NOCLOUD ? cx1 : sx[pos_old];
      nxyz[(size_t)(npoint - 1) * new_stride + 1] = NOCLOUD ? cy1 : sy[pos_old];
      nxyz[(size_t)(npoint - 1) * new_stride + 2] = NOCLOUD ? cz1 : sz[pos_old];
    }
  }
}

#ifndef MPX_FPS4_MIN_B
#define MPX_FPS4_MIN_B 768  // from this many environments on, the 4-wave culled FPS (A/B: tools/ab_build.sh)
#endif
constexpr int FPS_MAX_N = 8192;                         // 512 threads x 16 points (include/mpinets_hip.h says the same)
constexpr int FPS_MAX_LDS = 256 + 3 * FPS_MAX_N * 4;    // the cloud copy of the largest supported launch

static int opt_n_threads_log2(int n) {
  int l = 0;
  while ((2 << l) <= n && (2 << l) <= 512) ++l;
  return l;
}

// ---- verification hook: which kernel family serves mpx_fps / mpx_ball_query ---------------------------------------
// Variant 1 (default) = the fast kernels (wave-per-environment / Morton-culled FPS; wave-per-query / bucketed ball
// query), variant 0 = the plain kernels they are proven against (fps_kernel, ball_query_kernel).  Both produce the same
// indices bit for bit; the switch exists so that a test can run BOTH in one process on the same clouds
// (tests/test_gpu_soak.py).  Process-wide, read at every launch.
static std::atomic<int> g_variant[MPX_VARIANT_COUNT_] = {{1}, {1}, {1}};
MPX_EXPORT int mpx_set_variant(int what, int value) {
  MPX_REQUIRE(what >= 0 && what < MPX_VARIANT_COUNT_, "mpx_set_variant: unknown selector %d", what);
  MPX_REQUIRE(value == 0 || value == 1, "mpx_set_variant: value must be 0 (plain kernels) or 1 (default)");
  g_variant[what].store(value, std::memory_order_relaxed);
  // MPX_VARIANT_UNIT_QUEUE 0: streams not seen before get no unit-queue slot (as if all 256 were taken)
  if (what == MPX_VARIANT_UNIT_QUEUE) mpx_unit_queue_set_slots(value ? 256 : 0);
  return 0;
}
MPX_EXPORT int mpx_get_variant(int what) {
  return (what >= 0 && what < MPX_VARIANT_COUNT_) ? g_variant[what].load(std::memory_order_relaxed) : -1;
}

MPX_EXPORT int mpx_fps(const float *xyz, int B, int N, int stride, int npoint, int32_t *idx, float *new_xyz,
                       int new_stride, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N >= 1 && npoint >= 0 && stride >= 3, "mpx_fps: bad size");
  MPX_REQUIRE(N <= FPS_MAX_N, "mpx_fps: N = %d > %d unsupported", N, FPS_MAX_N);
  MPX_REQUIRE(new_xyz == nullptr || new_stride >= 3, "mpx_fps: new_stride < 3");
  if (B == 0 || npoint == 0) return 0;
  const int log2bs = opt_n_threads_log2(N);
  const int fast = g_variant[MPX_VARIANT_FPS].load(std::memory_order_relaxed);
  if (fast && N <= 512) {  // small cloud: one wave per environment, no LDS, no barrier
    dim3 g1(B), t1(64);
#define FPS_WAVE(P) hipLaunchKernelGGL(fps_wave_kernel<P>, g1, t1, 0, mpx_s(stream), xyz, N, stride, npoint, log2bs, idx, new_xyz, new_stride)
    switch ((N + 63) / 64) {
      case 1: FPS_WAVE(1); break;
      case 2: FPS_WAVE(2); break;
      case 3: case 4: FPS_WAVE(4); break;
      default: FPS_WAVE(8); break;
    }
#undef FPS_WAVE
    MPX_LAUNCH_CHECK("mpx_fps");
  }
  if (fast && N > 512) {  // (log2bs == 9 here: the key layout of fps_cull_kernel assumes it)
    // large clouds (the first module's 6272 points): the 4-wave form without the LDS copy of the cloud -- three
    // workgroups per CU instead of two (6.45 -> 6.11 ms at 8192 environments; with the LDS copy, two per CU: 7.10 ms)
    // (a few hundred environments or fewer are latency-bound, not throughput-bound: they keep the 8-wave form, whose pick
    // chain is shorter -- one planning problem: 0.69 vs 0.80 ms per step.  Same indices either way.)
    const int waves = (N > 16 * 256 && B >= MPX_FPS4_MIN_B) ? 4 : 8;
    const size_t lds_c = fpsc_lds_bytes(N, waves == 4);
    // (wave counts measured in round 3 at 8192 environments x 6272 points: 16 waves x 7 points per lane 8.31 ms, 8 x 13
    // 6.41 ms, 4 x 25 with the LDS cloud copy 7.10 ms, 4 x 25 without it 6.11 ms: the cross-wave reduction and the
    // barrier get cheaper with fewer waves, the per-wave pass longer; what pays is the third workgroup per CU.)
    dim3 gc(B), tc(64 * waves);
#define FPS_CULL(P, W)                                                                                           \
  do {                                                                                                           \
    if (lds_c > 64 * 1024) MPX_LDS_LIMIT_ONCE((fps_cull_kernel<P, W>), fpsc_lds_bytes(FPS_MAX_N), "mpx_fps");     \
    hipLaunchKernelGGL((fps_cull_kernel<P, W>), gc, tc, lds_c, mpx_s(stream), xyz, N, stride, npoint, idx, new_xyz, \
                       new_stride);                                                                              \
  } while (0)
    const int pts_c = (N + 64 * waves - 1) / (64 * waves);
    if (waves == 4) {
      if (pts_c <= 20) FPS_CULL(20, 4);
      else if (pts_c <= 25) FPS_CULL(25, 4);
      else FPS_CULL(32, 4);
    } else if (pts_c <= 2) FPS_CULL(2, 8);
    else if (pts_c <= 4) FPS_CULL(4, 8);
    else if (pts_c <= 6) FPS_CULL(6, 8);
    else if (pts_c <= 8) FPS_CULL(8, 8);
    else if (pts_c <= 10) FPS_CULL(10, 8);
    else if (pts_c <= 13) FPS_CULL(13, 8);
    else FPS_CULL(16, 8);
#undef FPS_CULL
    MPX_LAUNCH_CHECK("mpx_fps");
  }
  int block = ((N + 63) / 64) * 64;
  if (block > 512) block = 512;  // measured: 512 threads x 13 points beat 1024 x 7
  const int pts = (N + block - 1) / block;
  const size_t lds = 256 + (size_t)3 * N * sizeof(float);
  dim3 g(B), t(block);
#define FPS_LAUNCH(P)                                                                                     \
  do {                                                                                                    \
    if (lds > 64 * 1024) MPX_LDS_LIMIT_ONCE(fps_kernel<P>, FPS_MAX_LDS, "mpx_fps");                       \
    hipLaunchKernelGGL(fps_kernel<P>, g, t, lds, mpx_s(stream), xyz, N, stride, npoint, log2bs, idx,      \
                       new_xyz, new_stride);                                                              \
  } while (0)
  switch (pts) {
    case 1: FPS_LAUNCH(1); break;
    case 2: FPS_LAUNCH(2); break;
    case 3: FPS_LAUNCH(3); break;
    case 4: FPS_LAUNCH(4); break;
    case 5: FPS_LAUNCH(5); break;
    case 6: FPS_LAUNCH(6); break;
    case 7: FPS_LAUNCH(7); break;
    case 8: FPS_LAUNCH(8); break;
    case 9: case 10: FPS_LAUNCH(10); break;
    case 11: case 12: case 13: FPS_LAUNCH(13); break;
    case 14: case 15: case 16: FPS_LAUNCH(16); break;
    default:
      mpx_set_error("mpx_fps: %d points per thread unsupported (N=%d, block=%d)", pts, N, block);
      return 1;
  }
#undef FPS_LAUNCH
  MPX_LAUNCH_CHECK("mpx_fps");
}

// ---- ball query ------------------------------------------------------------------------------------
// One QUERY per lane (64 queries per wave, 256 per workgroup); the cloud is walked in index order
// with wave-uniform addresses, so every point arrives through the scalar cache as SGPR operands
// and costs one pass of ~9 VALU instructions for 64 queries.  A lane appends its hits to its own
// output row as they occur (index order is the loop order; few hits, scattered 4-byte stores);
// the padding of the remaining slots -- most of the 512-byte row -- is written cooperatively,
// one row at a time, fully coalesced.  Algorithmic HBM traffic: cloud read once per workgroup
// (L2-resident across an environment's workgroups) + npoint*nsample*4 bytes written.
// grid (ceil(npoint/256), B), block 256.
template <int STRIDE, bool ALIGNED64>
__global__ void __launch_bounds__(256)
    ball_query_kernel(const float *__restrict__ new_xyz, int new_stride, const float *__restrict__ xyz,
                      int stride_rt, int N, int npoint, float radius2, int nsample, int32_t *__restrict__ idx,
                      int32_t *__restrict__ cnt_out, int pad) {
  const int stride = STRIDE > 0 ? STRIDE : stride_rt;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < npoint;
  const int jc = live ? j : npoint - 1;
  const float *c = new_xyz + ((size_t)b * npoint + jc) * new_stride;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float *pts = xyz + (size_t)b * N * stride;  // block-uniform: scalar loads below
  int32_t *out = idx + ((size_t)b * npoint + jc) * nsample;
  int cnt = live ? 0 : nsample;  // dead lanes never take a hit
  int first = 0;
  auto test_point = [&](int k, float px, float py, float pz) {
    const float dx = cx - px, dy = cy - py, dz = cz - pz;
    const float d2 = mpx_sqdist(dx, dy, dz);
    if (d2 < radius2 && cnt < nsample) {
      if (cnt == 0) first = k;
      out[cnt] = k;
      ++cnt;
    }
  };
  constexpr int U = 8;  // points fetched per scalar-load batch
  int k = 0;
  if (STRIDE == 4 && ALIGNED64) {
    // slab rows: 4 points = one 64-byte scalar load (the launcher checked the alignment)
    // (hipcc turns a plain uniform 64-byte load into per-point vector loads here, so the scalar
    // loads and their wait are spelled out; the wait is inside the statement, section 5.7 rule 1.)
    typedef float f32x16s __attribute__((ext_vector_type(16)));
    for (; k + U <= N; k += U) {
      f32x16s a, c2;
      const float *src = pts + (size_t)k * 4;
      asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(a), "=&s"(c2)
                   : "s"(src)
                   : "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) test_point(k + u, a[4 * u], a[4 * u + 1], a[4 * u + 2]);
#pragma unroll
      for (int u = 0; u < 4; ++u) test_point(k + 4 + u, c2[4 * u], c2[4 * u + 1], c2[4 * u + 2]);
      if ((k & 63) == 56 && __all(cnt >= nsample)) break;
    }
  } else {
    for (; k + U <= N; k += U) {
      float p[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        p[u][0] = pts[(size_t)(k + u) * stride + 0];
        p[u][1] = pts[(size_t)(k + u) * stride + 1];
        p[u][2] = pts[(size_t)(k + u) * stride + 2];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) test_point(k + u, p[u][0], p[u][1], p[u][2]);
      if ((k & 63) == 56 && __all(cnt >= nsample)) break;  // every query of the wave is full
    }
  }
  if (!__all(cnt >= nsample))
    for (; k < N; ++k) test_point(k, pts[(size_t)k * stride + 0], pts[(size_t)k * stride + 1], pts[(size_t)k * stride + 2]);
  if (cnt_out && live) cnt_out[(size_t)b * npoint + j] = cnt < nsample ? cnt : nsample;
  if (!pad) {  // hit slots only: an empty row still gets its slot 0
    if (live && cnt == 0) out[0] = 0;
    return;
  }
  // cooperative, coalesced padding: row q of this wave gets `first_q` in slots [cnt_q, nsample)
  const int jw = j - lane;  // first query of this wave
  for (int q = 0; q < 64; ++q) {
    if (jw + q >= npoint) break;
    const int cq = __shfl(cnt, q), fq = __shfl(first, q);
    int32_t *row = idx + ((size_t)b * npoint + jw + q) * nsample;
    for (int l = cq + lane; l < nsample; l += 64) row[l] = fq;
  }
}

// ---- ball query of a SMALL cloud (N <= 512: the second set-abstraction module, 128 queries x 512 points, many hits) ----
// The kernel above gives a query to a lane, which appends its hits with scattered 4-byte stores: with ~60 hits per row
// that is store-issue bound (0.56 ms per step at 8192 environments).  Here a WAVE owns a query at a time and the cloud
// sits in its registers (lane l holds points l, l + 64, ...: 64 consecutive indices per register slot): a slot is one
// distance pass + one ballot; a hit's output position is the running count plus the hits in the lanes below it
// (v_mbcnt), so the row is written in index order with contiguous stores, and the padding by the same wave.
// Same arithmetic (mpx_sqdist of centre - point) and the same idx / cnt as ball_query_kernel.
template <int PTS>
__global__ void __launch_bounds__(256)
    ball_query_wave_kernel(const float *__restrict__ new_xyz, int new_stride, const float *__restrict__ xyz, int stride,
                           int N, int npoint, float radius2, int nsample, int32_t *__restrict__ idx,
                           int32_t *__restrict__ cnt_out, int qpw, int pad) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j0 = (blockIdx.x * 4 + wave) * qpw;
  if (j0 >= npoint) return;
  const float *pts = xyz + (size_t)b * N * stride;
  float px[PTS], py[PTS], pz[PTS];
#pragma unroll
  for (int i = 0; i < PTS; ++i) {
    const int k = lane + 64 * i;
    px[i] = py[i] = pz[i] = 0.0f;
    if (k < N) {
      px[i] = pts[(size_t)k * stride + 0];
      py[i] = pts[(size_t)k * stride + 1];
      pz[i] = pts[(size_t)k * stride + 2];
    }
  }
  const int j1 = min(j0 + qpw, npoint);
  for (int j = j0; j < j1; ++j) {  // (wave-uniform)
    const float *c = new_xyz + ((size_t)b * npoint + j) * new_stride;
    const float cx = c[0], cy = c[1], cz = c[2];
    int32_t *row = idx + ((size_t)b * npoint + j) * nsample;
    int base = 0, first = 0;
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
      if (64 * i < N && base < nsample) {  // (uniform)
        const float dx = cx - px[i], dy = cy - py[i], dz = cz - pz[i];
        const float d2 = mpx_sqdist(dx, dy, dz);
        const bool hit = lane + 64 * i < N && d2 < radius2;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (m) {
          const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (hit && pos < nsample) row[pos] = lane + 64 * i;
          if (base == 0) first = 64 * i + (int)__builtin_ctzll(m);
          base += (int)__builtin_popcountll(m);
        }
      }
    }
    const int cnt = base < nsample ? base : nsample;
    if (pad) {
      for (int l = cnt + lane; l < nsample; l += 64) row[l] = first;  // padding: the first hit (zeros when there is none)
    } else if (cnt == 0 && lane == 0) {
      row[0] = 0;  // (hit slots only: an empty row still gets its slot 0)
    }
    if (cnt_out && lane == 0) cnt_out[(size_t)b * npoint + j] = cnt;
  }
}

// ---- ball query through a column grid (exact; the large-cloud / small-radius case) ---------------------------
// The brute-force kernel above tests every (query, point) pair: 512 x 6272 for the first set-abstraction module,
// of which ~10 per query are hits (radius 5 cm).  Here one workgroup owns one environment: the cloud is bucketed
// in LDS into G x G vertical columns of side h >= radius over (x, y) (counting sort), a thread walks the 3 x 3
// columns around its query -- every point within the radius lies there -- and appends its hits to the output row;
// the rows are then sorted by point index inside the wave (bitonic network on the lanes), which restores exactly
// the reference's "first nsample hits in index order" (+ padding with the first hit).  A query that collects more
// than nsample hits needs the nsample SMALLEST indices: it is redone by one wave scanning the whole cloud in index
// order with ballot-ordered compaction (rare: a dense cluster such as the target gripper cloud).
// Same distance arithmetic, same strict comparison, so idx and cnt are bit-identical to the brute-force kernel.
constexpr int BQG = 48;            // columns per side
#ifndef MPX_BQG_THREADS
#define MPX_BQG_THREADS 1024
#endif
constexpr int BQG_THREADS = MPX_BQG_THREADS;  // 16 waves: the query walk is LDS-latency bound, two lanes share a query
constexpr int BQ_HC = 40;          // hits of a query kept in LDS (the rest of a long row goes through its global row)

__device__ __forceinline__ int bq_cell(float v, float origin, float inv_h) {
  const int c = (int)floorf((v - origin) * inv_h);
  return c < 0 ? 0 : (c >= BQG ? BQG - 1 : c);
}
__device__ __forceinline__ int sort64(int v, int lane) {  // ascending bitonic sort of one key per lane
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int o = __shfl_xor(v, j);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      v = (lower == up) ? min(v, o) : max(v, o);
    }
  return v;
}

// GS-lane groups of a wave each sort one short row (n <= GS keys) and write it out with its padding
template <int GS, class Fetch>
__device__ __forceinline__ void bq_sort_rows(int j, int jn_left, const unsigned short *qcnt, Fetch &&fetch, int32_t *rows,
                                             int nsample, int32_t *cnt_row, int lane, int pad) {
  const int g = lane / GS, hl = lane % GS, jj = j + g;
  const bool live = g < jn_left;
  const int nn = live ? qcnt[jj] : 0;
  int v = live ? fetch(jj, hl, nn) : 0x7FFFFFFF;
#pragma unroll
  for (int k = 2; k <= GS; k <<= 1)
#pragma unroll
    for (int s = k >> 1; s > 0; s >>= 1) {
      const int o = __shfl_xor(v, s);
      const bool up = (hl & k) == 0, lower = (hl & s) == 0;
      v = (lower == up) ? min(v, o) : max(v, o);
    }
  const int first = nn > 0 ? __shfl(v, g * GS) : 0;
  if (live) {
    int32_t *rr = rows + (size_t)jj * nsample;
    if (pad || hl < (nn > 0 ? nn : 1)) rr[hl] = hl < nn ? v : first;
    if (pad)
      for (int l = hl + GS; l < nsample; l += GS) rr[l] = first;
    if (cnt_row && hl == 0) cnt_row[jj] = nn;
  }
}

__global__ void __launch_bounds__(BQG_THREADS)
    ball_query_grid_kernel(const float *__restrict__ new_xyz, int new_stride, const float *__restrict__ xyz, int stride,
                           int N, int npoint, float radius2, float inv_h, int nsample, int32_t *__restrict__ idx,
                           int32_t *__restrict__ cnt_out, int pad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *sp = reinterpret_cast<float4 *>(smem);                                    // [N] (x, y, z, point id) by column
  int *ccount = reinterpret_cast<int *>(sp + N);                                    // [G*G] counts, then cursors
  unsigned short *cstart = reinterpret_cast<unsigned short *>(ccount + BQG * BQG);  // [G*G + 1]
  unsigned short *qcnt = cstart + BQG * BQG + 2;                                    // [npoint] hits per query
  unsigned short *ovf = qcnt + ((npoint + 1) & ~1);                                 // [npoint] overflowing queries
  unsigned short *hbuf = ovf + ((npoint + 1) & ~1);                                 // [npoint][BQ_HC] first hits of a row
  int *cnt_s = reinterpret_cast<int *>(hbuf + (size_t)npoint * BQ_HC);              // [npoint] hits found so far (atomic)
  __shared__ float red[2 * (BQG_THREADS / 64)];
  __shared__ int scan_s[BQG_THREADS / 64];
  __shared__ int n_ovf;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = xyz + (size_t)b * N * stride;
  const float *ctr = new_xyz + (size_t)b * npoint * new_stride;
  int32_t *rows = idx + (size_t)b * npoint * nsample;

  // ---- cloud -> registers, lower corner of its (x, y) bounding box
  constexpr int PT = 8192 / BQG_THREADS;  // points per thread (N <= 8192)
  float px[PT], py[PT], pz[PT];
  float mnx = __builtin_inff(), mny = __builtin_inff();
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int k = tid + i * BQG_THREADS;
    px[i] = py[i] = pz[i] = 0.0f;
    if (k < N) {
      px[i] = pts[(size_t)k * stride], py[i] = pts[(size_t)k * stride + 1], pz[i] = pts[(size_t)k * stride + 2];
      mnx = fminf(mnx, px[i]), mny = fminf(mny, py[i]);
    }
  }
  for (int i = tid; i < BQG * BQG; i += BQG_THREADS) ccount[i] = 0;
  for (int i = tid; i < npoint; i += BQG_THREADS) cnt_s[i] = 0;
  if (tid == 0) n_ovf = 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mnx = fminf(mnx, __shfl_xor(mnx, o)), mny = fminf(mny, __shfl_xor(mny, o));
  if (lane == 0) red[2 * wave] = mnx, red[2 * wave + 1] = mny;
  __syncthreads();
  float ox = red[0], oy = red[1];
#pragma unroll
  for (int w = 1; w < BQG_THREADS / 64; ++w) ox = fminf(ox, red[2 * w]), oy = fminf(oy, red[2 * w + 1]);

  // ---- counting sort by column: LDS receives (x, y, z, point id) IN BUCKET ORDER as one 16-byte record, so a candidate
  // costs ONE ds_read_b128.  (The walk below reads per-lane random bucket slots: as three ds_read_b32 + a 16-bit id read
  // it serialised on LDS bank conflicts -- 67-96 k of the kernel's 165 k cycles per environment by s_memtime, and not an
  // imbalance: the same candidates dealt evenly over the threads took as long.)
  int cell[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    cell[i] = bq_cell(px[i], ox, inv_h) * BQG + bq_cell(py[i], oy, inv_h);
    if (tid + i * BQG_THREADS < N) atomicAdd(&ccount[cell[i]], 1);
  }
  __syncthreads();
  {
    constexpr int PER = (BQG * BQG + BQG_THREADS - 1) / BQG_THREADS;
    int local[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid * PER + i;
      local[i] = c < BQG * BQG ? ccount[c] : 0;
      sum += local[i];
    }
    int inc = sum;  // inclusive scan over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) scan_s[wave] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < wave; ++w) base += scan_s[w];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid * PER + i;
      if (c < BQG * BQG) {
        cstart[c] = (unsigned short)base;
        ccount[c] = base;  // running cursor for the scatter
        base += local[i];
      }
    }
    if (tid == BQG_THREADS - 1) cstart[BQG * BQG] = (unsigned short)N;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int k = tid + i * BQG_THREADS;
    if (k < N) {
      const int at = atomicAdd(&ccount[cell[i]], 1);
      sp[at] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
    }
  }
  __syncthreads();

  auto col_range = [&](float cx, float cy, int &x0, int &x1, int &y0, int &y1) __attribute__((always_inline)) {
    const int ix = (int)floorf((cx - ox) * inv_h), iy = (int)floorf((cy - oy) * inv_h);
    x0 = min(max(ix - 1, 0), BQG - 1), x1 = min(max(ix + 1, 0), BQG - 1);
    y0 = min(max(iy - 1, 0), BQG - 1), y1 = min(max(iy + 1, 0), BQG - 1);
  };

  // ---- TWO lanes per query (the walk is a chain of dependent LDS reads: twice the lanes, twice the reads in flight):
  // each takes one half of every column range of the 3 x 3 neighbourhood; hits are appended unsorted to the query's row
  // at slots handed out by an LDS counter (the rows are sorted by point index afterwards, so the order of arrival is
  // irrelevant: idx / cnt stay bit-identical)
  for (int w = tid; w < 2 * npoint; w += BQG_THREADS) {
    const int j = w >> 1, h = w & 1;
    const float cx = ctr[(size_t)j * new_stride], cy = ctr[(size_t)j * new_stride + 1], cz = ctr[(size_t)j * new_stride + 2];
    int x0, x1, y0, y1;
    col_range(cx, cy, x0, x1, y0, y1);
    int32_t *out = rows + (size_t)j * nsample;
    auto hit = [&](int k) __attribute__((always_inline)) {
      const int slot = atomicAdd(&cnt_s[j], 1);
      if (slot < BQ_HC) hbuf[j * BQ_HC + slot] = (unsigned short)k;
      else if (slot < nsample) out[slot] = k;
    };
    for (int gx = x0; gx <= x1; ++gx) {
      // columns y0..y1 of one x are adjacent in the sorted order: one contiguous range, halved between the two lanes
      const int r0 = cstart[gx * BQG + y0], r1 = cstart[gx * BQG + y1 + 1], mid = r0 + ((r1 - r0 + 1) >> 1);
      const int e0 = h ? mid : r0, e1 = h ? r1 : mid;
      int e = e0;
      for (; e + 4 <= e1; e += 4) {  // four candidates in flight
        float d2[4];
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 p = sp[e + u];
          const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
          d2[u] = mpx_sqdist(dx, dy, dz);
          kk[u] = __float_as_int(p.w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (d2[u] < radius2) hit(kk[u]);
      }
      for (; e < e1; ++e) {
        const float4 p = sp[e];
        const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
        if (mpx_sqdist(dx, dy, dz) < radius2) hit(__float_as_int(p.w));
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int j = tid; j < npoint; j += BQG_THREADS) {
    const int c = cnt_s[j];
    if (c > nsample) ovf[atomicAdd(&n_ovf, 1)] = (unsigned short)j;  // needs the nsample SMALLEST indices: redone below
    qcnt[j] = (unsigned short)(c < nsample ? c : nsample);
  }
  __threadfence_block();
  __syncthreads();

  // ---- queries with more than nsample hits: whole cloud in index order, ballot-ordered compaction (sorted by construction)
  const int novf = n_ovf;
  for (int o = wave; o < novf; o += BQG_THREADS / 64) {
    const int j = ovf[o];
    const float cx = ctr[(size_t)j * new_stride], cy = ctr[(size_t)j * new_stride + 1], cz = ctr[(size_t)j * new_stride + 2];
    int32_t *out = rows + (size_t)j * nsample;
    int have = 0;
    for (int k0 = 0; k0 < N && have < nsample; k0 += 64) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < N) {  // (index order: from global memory -- LDS holds the cloud in bucket order)
        const float dx = cx - pts[(size_t)k * stride], dy = cy - pts[(size_t)k * stride + 1], dz = cz - pts[(size_t)k * stride + 2];
        hit = mpx_sqdist(dx, dy, dz) < radius2;
      }
      const unsigned long long m = __ballot(hit);
      const int pos = have + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < nsample) {
        if (pos < BQ_HC) hbuf[j * BQ_HC + pos] = (unsigned short)k;
        else out[pos] = k;
      }
      have += __popcll(m);
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- per row: sort the hits by point index, pad with the first one, write the count
  auto fetch = [&](int j, int e, int n) __attribute__((always_inline)) {  // element e of row j (or +inf)
    if (e >= n) return 0x7FFFFFFF;
    return e < BQ_HC ? (int)hbuf[j * BQ_HC + e] : rows[(size_t)j * nsample + e];
  };
  // rows are dealt to the waves in chunks (a multiple of 4: short rows are sorted four at a time) sized so that every
  // wave of the workgroup gets some: 512 rows on 16 waves = 32 each
  const int rpw = min(64, max(4, ((npoint + BQG_THREADS / 64 - 1) / (BQG_THREADS / 64) + 3) & ~3));
  for (int j0 = wave * rpw; j0 < npoint; j0 += rpw * (BQG_THREADS / 64)) {
    const int jn = min(rpw, npoint - j0);
    int q = 0;
    while (q < jn) {
      const int j = j0 + q, n = qcnt[j];
      const int n2 = q + 1 < jn ? qcnt[j + 1] : 0;
      int m4 = max(n, n2);
      if (q + 2 < jn) m4 = max(m4, (int)qcnt[j + 2]);
      if (q + 3 < jn) m4 = max(m4, (int)qcnt[j + 3]);
      int32_t *crow = cnt_out ? cnt_out + (size_t)b * npoint : nullptr;
      if (m4 <= 16) {  // four short rows at once, one per 16 lanes
        bq_sort_rows<16>(j, jn - q, qcnt, fetch, rows, nsample, crow, lane, pad);
        q += 4;
        continue;
      }
      if (n <= 32 && n2 <= 32) {  // two rows, one per half-wave
        bq_sort_rows<32>(j, jn - q, qcnt, fetch, rows, nsample, crow, lane, pad);
        q += 2;
        continue;
      }
      int32_t *row = rows + (size_t)j * nsample;
      int a = fetch(j, lane, n), bb = 0x7FFFFFFF;
      if (n > 64) {  // two keys per lane: elements lane and lane + 64 of a 128-key network
        bb = fetch(j, lane + 64, n);
#pragma unroll
        for (int k = 2; k <= 128; k <<= 1)
#pragma unroll
          for (int s = k >> 1; s > 0; s >>= 1) {
            if (s == 64) {
              const int lo = min(a, bb), hi = max(a, bb);
              a = lo, bb = hi;
            } else {
              const int oa = __shfl_xor(a, s), ob = __shfl_xor(bb, s);
              const bool lower = (lane & s) == 0;
              const bool upa = (lane & k) == 0, upb = ((lane + 64) & k) == 0;
              a = (lower == upa) ? min(a, oa) : max(a, oa);
              bb = (lower == upb) ? min(bb, ob) : max(bb, ob);
            }
          }
      } else if (n > 1) {
        a = sort64(a, lane);
      }
      const int first = n > 0 ? __shfl(a, 0) : 0;
      const int n_write = pad ? nsample : (n > 0 ? n : 1);  // (pad == 0: the hit slots only; an empty row still gets its slot 0)
      for (int l = lane; l < n_write; l += 64) {
        const int v = l < 64 ? a : bb;
        row[l] = l < n ? v : first;
      }
      if (cnt_out && lane == 0) cnt_out[(size_t)b * npoint + j] = n;
      ++q;
    }
  }
  __syncthreads();
}

static int ball_query_impl(const float *new_xyz, int new_stride, const float *xyz, int stride, int B, int N, int npoint,
                           float radius, int nsample, int32_t *idx, int32_t *cnt, int pad, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N >= 0 && npoint >= 0 && nsample >= 0, "mpx_ball_query: negative size");
  MPX_REQUIRE(stride >= 3 && new_stride >= 3, "mpx_ball_query: stride < 3");
  if (B == 0 || npoint == 0 || nsample == 0) return 0;
  if (B > MPX_GRID_Y) {  // more environments than one launch's gridDim.y: slabs
    for (int64_t b0 = 0; b0 < B; b0 += MPX_GRID_Y)
      if (int rc = ball_query_impl(new_xyz + b0 * npoint * new_stride, new_stride, xyz + b0 * N * stride, stride,
                                   (int)(B - b0 < MPX_GRID_Y ? B - b0 : MPX_GRID_Y), N, npoint, radius, nsample,
                                   idx + b0 * npoint * nsample, cnt ? cnt + b0 * npoint : nullptr, pad, stream))
        return rc;
    return 0;
  }
  const float r2 = radius * radius;  // float product, like the reference kernel
  // large cloud, small radius (the first set-abstraction module): bucketed search, bit-identical output
  const int fast = g_variant[MPX_VARIANT_BALL_QUERY].load(std::memory_order_relaxed);
  if (fast && N >= 2048 && N <= 8192 && nsample <= 128 && nsample > BQ_HC && npoint <= 4096 && radius > 0.0f &&
      radius * BQG < 4.0f) {  // columns of side ~radius must still resolve the scene (48 x radius < 4 m)
    const size_t lds = (size_t)4 * N * 4 + (size_t)BQG * BQG * 4 + (size_t)(BQG * BQG + 2) * 2 +
                       (size_t)((npoint + 1) & ~1) * 2 * 2 + (size_t)npoint * BQ_HC * 2 + (size_t)npoint * 4;
    if (lds <= 158 * 1024) {
      MPX_LDS_LIMIT_ONCE(ball_query_grid_kernel, 158 * 1024, "mpx_ball_query");
      const float inv_h = 1.0f / (radius * 1.0001f);
      hipLaunchKernelGGL(ball_query_grid_kernel, dim3(B), dim3(BQG_THREADS), lds, mpx_s(stream), new_xyz, new_stride, xyz,
                         stride, N, npoint, r2, inv_h, nsample, idx, cnt, pad);
      MPX_LAUNCH_CHECK("mpx_ball_query");
    }
  }
  if (fast && N >= 1 && N <= 512) {  // small cloud: a wave per query, cloud in registers, rows written in order
    const int qpw = npoint >= 64 ? 16 : 4;  // queries per wave (the cloud load is amortised over them)
    dim3 gw(cdiv(npoint, 4 * qpw), B), tw(256);
#define BQ_WAVE(P)                                                                                                   \
  hipLaunchKernelGGL(ball_query_wave_kernel<P>, gw, tw, 0, mpx_s(stream), new_xyz, new_stride, xyz, stride, N, npoint, r2, \
                     nsample, idx, cnt, qpw, pad)
    if (N <= 128) BQ_WAVE(2);
    else if (N <= 256) BQ_WAVE(4);
    else BQ_WAVE(8);
#undef BQ_WAVE
    MPX_LAUNCH_CHECK("mpx_ball_query");
  }
  dim3 g(cdiv(npoint, 256), B), t(256);
  const bool al64 = stride == 4 && ((uintptr_t)xyz & 63) == 0 && N % 4 == 0;
  if (al64)
    hipLaunchKernelGGL((ball_query_kernel<4, true>), g, t, 0, mpx_s(stream), new_xyz, new_stride, xyz, stride, N, npoint,
                       r2, nsample, idx, cnt, pad);
  else if (stride == 4)
    hipLaunchKernelGGL((ball_query_kernel<4, false>), g, t, 0, mpx_s(stream), new_xyz, new_stride, xyz, stride, N, npoint,
                       r2, nsample, idx, cnt, pad);
  else if (stride == 3)
    hipLaunchKernelGGL((ball_query_kernel<3, false>), g, t, 0, mpx_s(stream), new_xyz, new_stride, xyz, stride, N, npoint,
                       r2, nsample, idx, cnt, pad);
  else
    hipLaunchKernelGGL((ball_query_kernel<0, false>), g, t, 0, mpx_s(stream), new_xyz, new_stride, xyz, stride, N, npoint,
                       r2, nsample, idx, cnt, pad);
  MPX_LAUNCH_CHECK("mpx_ball_query");
}

MPX_EXPORT int mpx_ball_query(const float *new_xyz, int new_stride, const float *xyz, int stride, int B, int N,
                              int npoint, float radius, int nsample, int32_t *idx, int32_t *cnt,
                              mpx_stream_t stream) {
  return ball_query_impl(new_xyz, new_stride, xyz, stride, B, N, npoint, radius, nsample, idx, cnt, 1, stream);
}
// the hit slots only: idx[b, j, 0 .. max(cnt, 1)) are written, the padding slots are LEFT UNTOUCHED -- for consumers that
// take the hit count and never look past it (the fused grouped-MLP kernels with `cnt`): the first module's rows are 12 %
// hits on the bench scenes, so 1.9 of the 2.15 GB of index writes per step go away.  cnt is required.
MPX_EXPORT int mpx_ball_query_hits(const float *new_xyz, int new_stride, const float *xyz, int stride, int B, int N,
                                   int npoint, float radius, int nsample, int32_t *idx, int32_t *cnt,
                                   mpx_stream_t stream) {
  MPX_REQUIRE(cnt != nullptr, "mpx_ball_query_hits: the hit counts are required (they say which slots were written)");
  // (the grouped-MLP kernels honour `cnt` up to 256 slots per neighbourhood; above that mpx_sa_mlp walks every slot, and
  // would read the slots this entry point leaves unwritten)
  MPX_REQUIRE(nsample <= 256, "mpx_ball_query_hits: nsample = %d > 256 (use mpx_ball_query: full rows)", nsample);
  return ball_query_impl(new_xyz, new_stride, xyz, stride, B, N, npoint, radius, nsample, idx, cnt, 0, stream);
}

// ---- QueryAndGroup, materialised (API parity with the reference's unfused path) ---------------------
// out [B, 3+C, npoint, nsample]; thread per (j,l) pair, channel loop writes coalesced planes.
__global__ void __launch_bounds__(256)
    group_points_kernel(const float *__restrict__ xyz, int stride, const float *__restrict__ new_xyz,
                        int new_stride, const float *__restrict__ feat, int feat_stride, int C,
                        const int32_t *__restrict__ idx, int N, int npoint, int nsample, float *__restrict__ out) {
  const int b = blockIdx.y;
  const size_t plane = (size_t)npoint * nsample;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= plane) return;
  const int j = (int)(e / nsample);
  const int k = idx[(size_t)b * plane + e];
  const float *p = xyz + ((size_t)b * N + k) * stride;
  const float *c = new_xyz + ((size_t)b * npoint + j) * new_stride;
  float *o = out + (size_t)b * (3 + C) * plane + e;
  o[0 * plane] = p[0] - c[0];
  o[1 * plane] = p[1] - c[1];
  o[2 * plane] = p[2] - c[2];
  if (C > 0) {
    const float *f = feat + ((size_t)b * N + k) * feat_stride;
    for (int ch = 0; ch < C; ++ch) o[(size_t)(3 + ch) * plane] = f[ch];
  }
}

MPX_EXPORT int mpx_group_points(const float *xyz, int stride, const float *new_xyz, int new_stride,
                                const float *feat, int feat_stride, int C, const int32_t *idx, int B, int N,
                                int npoint, int nsample, float *out, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && B <= 65535 && C >= 0, "mpx_group_points: bad size");
  if (B == 0 || npoint == 0 || nsample == 0) return 0;
  hipLaunchKernelGGL(group_points_kernel, dim3(cdiv((int64_t)npoint * nsample, 256), B), dim3(256), 0,
                     mpx_s(stream), xyz, stride, new_xyz, new_stride, feat, feat_stride, C, idx, N, npoint,
                     nsample, out);
  MPX_LAUNCH_CHECK("mpx_group_points");
}

// ---- queries sorted by the number of rows they contribute (counting sort, <= 64 bins) ------------------
// rows = distinct neighbours rounded up to a multiple of 4 (what the packed SA kernels evaluate);
// bin = rows/4 - 1.  Decreasing order: the longest queries go first.
constexpr int SORT_BINS = 64;
__device__ __forceinline__ int sort_bin(int c, int nsample) {
  const int cc = c <= 0 ? 1 : (c > nsample ? nsample : c);
  return ((cc + 3) >> 2) - 1;
}

__global__ void __launch_bounds__(256)
    tile_hist_kernel(const int32_t *__restrict__ cnt, int64_t n, int nsample, int32_t *__restrict__ hist) {
  __shared__ int lh[SORT_BINS];
  if (threadIdx.x < SORT_BINS) lh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(&lh[sort_bin(cnt[i], nsample)], 1);
  __syncthreads();
  if (threadIdx.x < SORT_BINS && lh[threadIdx.x]) atomicAdd(hist + threadIdx.x, lh[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
    tile_scatter_kernel(const int32_t *__restrict__ cnt, int64_t n, int nsample, const int32_t *__restrict__ hist,
                        int32_t *__restrict__ cursor, int32_t *__restrict__ order) {
  // block-local ranking in LDS, then ONE global atomic per (block, bin) to reserve a range
  __shared__ int lh[SORT_BINS], lbase[SORT_BINS];
  if (threadIdx.x < SORT_BINS) lh[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int bin = 0, rank = 0;
  if (i < n) {
    bin = sort_bin(cnt[i], nsample);
    rank = atomicAdd(&lh[bin], 1);
  }
  __syncthreads();
  if (threadIdx.x < SORT_BINS) {
    int base = 0;  // bins in decreasing row count
    for (int b = SORT_BINS - 1; b > (int)threadIdx.x; --b) base += hist[b];
    lbase[threadIdx.x] = base + (lh[threadIdx.x] ? atomicAdd(cursor + threadIdx.x, lh[threadIdx.x]) : 0);
  }
  __syncthreads();
  if (i < n) order[lbase[bin] + rank] = (int32_t)i;
}

MPX_EXPORT int mpx_sort_queries(const int32_t *cnt, int64_t n, int nsample, int32_t *order, int32_t *scratch,
                                mpx_stream_t stream) {
  MPX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31), "mpx_sort_queries: bad n");
  MPX_REQUIRE(nsample > 0 && nsample <= 4 * SORT_BINS, "mpx_sort_queries: nsample must be in (0, %d]", 4 * SORT_BINS);
  if (n == 0) return 0;
  hipError_t e = hipMemsetAsync(scratch, 0, 2 * SORT_BINS * sizeof(int32_t), mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_sort_queries: memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(tile_hist_kernel, dim3(cdiv(n, 256)), dim3(256), 0, mpx_s(stream), cnt, n, nsample, scratch);
  hipLaunchKernelGGL(tile_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, mpx_s(stream), cnt, n, nsample, scratch,
                     scratch + SORT_BINS, order);
  MPX_LAUNCH_CHECK("mpx_sort_queries");
}
