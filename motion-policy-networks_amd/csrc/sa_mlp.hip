// sa_mlp.hip -- fused QueryAndGroup + shared MLP (3 x [1x1 conv + ReLU]) + neighbourhood max-pool,
// i.e. the body of pointnet2_ops' PointnetSAModule.forward as the reference configures it
// (/root/reference/mpinets/model.py:366-382: SA1 mlp [1(+3),64,64,64], SA2 mlp [64(+3),128,128,256],
// bn=False, nsample 128).  The reference materialises the grouped tensor [B,3+C,npoint,nsample] and
// every activation in HBM (16.8 MB per environment per SA1 layer); here nothing but the pooled
// [npoint, c3] rows is written.
//
// CDNA4 mapping (exact-fp32 matrix cores, v_mfma_f32_32x32x2_f32):
//   * one wave owns one query point and walks its neighbourhood 32 points at a time;
//   * layers 1 and 2 are computed as  H^T = W . X^T  (A operand = weights, B operand = activations):
//     the 32x32 result tile then has the POINT on the lane axis (col = lane&31) and the output
//     channels on the register axis (row = (r&3) + 8*(r>>2) + 4*(lane>>5)) -- which is already the
//     B-operand layout of the next layer if the k-steps walk the input channels in that same
//     (register, half) order.  So activations never leave registers, never move across lanes and
//     need no LDS: k-step t of a layer whose input came from tile `it` register `r` simply feeds
//     acc[it][r].  The weight stream is packed on the host side (mpx_sa_pack_weights) in exactly the
//     order the k-steps consume it: one float per lane per MFMA, fetched 4 steps at a time (16 B).
//   * the last layer flips roles (A = activations, B = weights) so that the POINTS land on the
//     register axis: the max-pool over the neighbourhood is then an in-lane max over 16 registers,
//     one cross-half exchange at the very end, and bias + ReLU are applied after pooling
//     (max and x -> relu(x + b) commute).
// The kernel is bound by the fp32 MFMA pipe (64 cycles per 32x32x2 on each SIMD; 157 TFLOP/s chip
// peak): per neighbourhood tile SA1 issues 132 and SA2 904 MFMAs while loading 2 KB of weights per
// 8 MFMAs from L1/L2 -- 16 B/clk/CU.  Summation order differs from a plain dot product (k walks the
// register order above), which is the only numerical difference to the oracle.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PAD_CH (-1)
constexpr int SA_MAX_NSAMPLE = 256;  // slots per neighbourhood the packed kernels index their row map for
// queries per unit of the large launches (A/B builds: narrow module 16 / 32: 10.38 / 10.14 ms -- the tail of a unit's last
// tile is 2.7 instead of 5.5 % of its rows; wide module 8 / 16: 46.6 / 46.8 ms -- its units are already 16 tiles)
#ifndef MPX_SA1_Q
#define MPX_SA1_Q 32
#endif
#ifndef MPX_SA2_Q
#define MPX_SA2_Q 8
#endif
#ifndef MPX_SA1_GR
#define MPX_SA1_GR 2
#endif
#ifndef MPX_SA_SPREAD
#define MPX_SA_SPREAD 1
#endif
// narrow module: pool through LDS float-max atomics into a [Q][C3] block that is written out once per unit, instead of
// merging the row groups in registers with a flush at every query boundary (round 6: 10.06 -> 9.61 ms at 8192 environments;
// 0 = the register merge, which also serves neighbourhoods of more than 128 slots and unaligned output rows)
#ifndef MPX_SA1_LDSPOOL
#define MPX_SA1_LDSPOOL 1
#endif

template <int CF, int C1, int C2, int C3>
struct SaCfg {
  static_assert(CF == 1 || CF % 2 == 0, "feature channels must be 1 or even");
  static_assert(C1 % 32 == 0 && C2 % 32 == 0 && C3 % 32 == 0, "layer widths must be multiples of 32");
  static constexpr int CIN = 3 + CF;
  static constexpr int KS0 = (CF == 1) ? 2 : 2 + CF / 2;  // k-steps of layer 1
  static constexpr int KS1 = C1 / 2, KS2 = C2 / 2;        // k-steps of layers 2, 3
  static constexpr int OT1 = C1 / 32, OT2 = C2 / 32, OT3 = C3 / 32;
  static constexpr int S1 = KS0 * OT1, S2 = KS1 * OT2, S3 = OT3 * KS2;  // MFMA steps per layer
  static_assert(S1 % 4 == 0 && S2 % 4 == 0 && S3 % 4 == 0, "steps are fetched four at a time");
  static constexpr int64_t W1_OFF = 0, W2_OFF = (int64_t)S1 * 64, W3_OFF = W2_OFF + (int64_t)S2 * 64;
  static constexpr int64_t B1_OFF = W3_OFF + (int64_t)S3 * 64, B2_OFF = B1_OFF + C1, B3_OFF = B2_OFF + C2;
  static constexpr int64_t TOTAL = B3_OFF + C3;

  // input channel consumed by k-step t on lane-half h, for each layer
  __host__ __device__ static int chan0(int t, int h) {
    if (CF == 1) return 2 * t + h;  // (dx,dy), (dz,label)
    if (t == 0) return h;           // (dx,dy)
    if (t == 1) return h ? PAD_CH : 2;  // (dz, -)
    return 3 + h * (CF / 2) + (t - 2);  // each half streams a contiguous half of the feature row
  }
  __host__ __device__ static int chan_tile(int t, int h) {
    const int it = t >> 4, r = t & 15;
    return it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
  }
};

// ---- weight packing ----------------------------------------------------------------------------------
template <class Cfg>
__global__ void __launch_bounds__(256)
    sa_pack_kernel(const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
                   const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3,
                   int c1, int c2, int c3, float *__restrict__ wpack) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= Cfg::TOTAL) return;
  float v;
  if (e >= Cfg::B1_OFF) {
    const int64_t i = e - Cfg::B1_OFF;
    v = i < c1 ? b1[i] : (i < c1 + c2 ? b2[i - c1] : b3[i - c1 - c2]);
  } else {
    // element (group g of 4 steps, lane, step-in-group)
    int layer;
    int64_t r = e;
    if (r >= Cfg::W3_OFF) { layer = 3; r -= Cfg::W3_OFF; }
    else if (r >= Cfg::W2_OFF) { layer = 2; r -= Cfg::W2_OFF; }
    else { layer = 1; }
    const int s = (int)(r / 256) * 4 + (int)(r & 3);
    const int lane = (int)((r >> 2) & 63);
    const int h = lane >> 5, o32 = lane & 31;
    int t, ot, in, cin;
    const float *w;
    if (layer == 1) { t = s / Cfg::OT1; ot = s % Cfg::OT1; in = Cfg::chan0(t, h); cin = Cfg::CIN; w = w1; }
    else if (layer == 2) { t = s / Cfg::OT2; ot = s % Cfg::OT2; in = Cfg::chan_tile(t, h); cin = c1; w = w2; }
    else { ot = s / Cfg::KS2; t = s % Cfg::KS2; in = Cfg::chan_tile(t, h); cin = c2; w = w3; }
    v = in == PAD_CH ? 0.0f : w[(size_t)(ot * 32 + o32) * cin + in];
  }
  wpack[e] = v;
}

// ---- the fused kernel ----------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load: wave-uniform descriptor + scalar byte offset + per-lane byte offset.  All
// address arithmetic stays on the scalar unit (no 64-bit VGPR pointers to keep alive or spill).
__device__ __forceinline__ float4 bload16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

__device__ __forceinline__ f32x16 bias_tile(__amdgpu_buffer_rsrc_t rsrc, int bias_off_bytes, int ot, int half) {
  // register r of a tile holds channel ot*32 + (r&3) + 8*(r>>2) + 4*half
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = bload16(rsrc, half * 16, bias_off_bytes + (ot * 32 + 8 * g) * 4);
    v[4 * g + 0] = q.x;
    v[4 * g + 1] = q.y;
    v[4 * g + 2] = q.z;
    v[4 * g + 3] = q.w;
  }
  return v;
}

__device__ __forceinline__ float comp(const float4 &q, int i) {
  return i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w));
}


// Streams NG float4 weight groups (one per 4 MFMA steps) through a two-deep register ring:
// chunk c+1 is in flight while chunk c feeds the matrix pipe.  The empty asm statements are
// compiler barriers for memory operations only -- without them hipcc hoists every load of the
// fully unrolled layer to its top and spills.
template <int NG, int CH = 4, class F>
__device__ __forceinline__ void stream_weights(__amdgpu_buffer_rsrc_t rsrc, int voff, int base, F &&body) {
  constexpr int NC = (NG + CH - 1) / CH;
  float4 buf[2][CH];
#pragma unroll
  for (int u = 0; u < CH; ++u)
    if (u < NG) buf[0][u] = bload16(rsrc, voff, base + u * 1024);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c + 1 < NC) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if ((c + 1) * CH + u < NG) buf[(c + 1) & 1][u] = bload16(rsrc, voff, base + ((c + 1) * CH + u) * 1024);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (c * CH + u < NG) body(c * CH + u, buf[c & 1][u]);
  }
}

template <int CF, int C1, int C2, int C3>
__global__ void __launch_bounds__(256, 2)
    sa_mlp_kernel(const float *__restrict__ xyz, int stride, const float *__restrict__ new_xyz, int new_stride,
                  const float *__restrict__ feat, int feat_stride, const int32_t *__restrict__ idx,
                  int64_t n_query, int N, int npoint, int nsample, const float *__restrict__ wpack,
                  float *__restrict__ out, int out_stride) {
  using Cfg = SaCfg<CF, C1, C2, C3>;
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, col = lane & 31;
  const int64_t qid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qid >= n_query) return;  // wave-uniform
  const int64_t b = qid / npoint;

  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wpack), 0, (int)(Cfg::TOTAL * 4), 0x00020000);
  const int wvoff = lane * 16;
  const float *bias3 = wpack + Cfg::B3_OFF;

  const float *ctr = new_xyz + qid * new_stride;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int32_t *nbr = idx + qid * nsample;
  const float *cloud = xyz + b * N * (int64_t)stride;
  const float *fbase = feat + b * N * (int64_t)feat_stride;

  float omax[Cfg::OT3];
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) omax[ot] = -__builtin_inff();

  for (int rt = 0; rt < nsample; rt += 32) {
    // Loop-invariant buffer offsets are laundered through an empty asm each iteration: otherwise LICM
    // hoists the (invariant) bias and first-chunk weight loads out of the loop and they are spilled.
    int w1o = (int)Cfg::W1_OFF * 4, w2o = (int)Cfg::W2_OFF * 4, w3o = (int)Cfg::W3_OFF * 4;
    int b1o = (int)Cfg::B1_OFF * 4, b2o = (int)Cfg::B2_OFF * 4;
    asm volatile("" : "+s"(w1o), "+s"(w2o), "+s"(w3o), "+s"(b1o), "+s"(b2o));
    const int k = nbr[rt + col];
    const float *p = cloud + (int64_t)k * stride;
    const float *f = fbase + (int64_t)k * feat_stride;

    // ---- layer-1 B operands: this lane's half of its point's input vector ---------------------------
    float x0[Cfg::KS0];
    {
      const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
      x0[0] = half ? dy : dx;
      if (CF == 1) {
        x0[1] = half ? f[0] : dz;
      } else {
        x0[1] = half ? 0.0f : dz;
        const float4 *fr = reinterpret_cast<const float4 *>(f + half * (CF / 2));
#pragma unroll
        for (int i = 0; i < CF / 8; ++i) {
          const float4 v = fr[i];
          x0[2 + 4 * i + 0] = v.x;
          x0[2 + 4 * i + 1] = v.y;
          x0[2 + 4 * i + 2] = v.z;
          x0[2 + 4 * i + 3] = v.w;
        }
      }
    }

    // ---- layer 1: H1^T = W1 . X^T --------------------------------------------------------------------
    f32x16 a1[Cfg::OT1];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT1; ++ot) a1[ot] = bias_tile(wrsrc, b1o, ot, half);
    stream_weights<Cfg::S1 / 4>(wrsrc, wvoff, w1o, [&](int g, const float4 &w) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = 4 * g + u, t = s / Cfg::OT1, ot = s % Cfg::OT1;
        a1[ot] = mfma32(comp(w, u), x0[t], a1[ot]);
      }
    });
#pragma unroll
    for (int ot = 0; ot < Cfg::OT1; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[ot][r] = fmaxf(a1[ot][r], 0.0f);

    // ---- layer 2: H2^T = W2 . H1^T ---------------------------------------------------------------------
    f32x16 a2[Cfg::OT2];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT2; ++ot) a2[ot] = bias_tile(wrsrc, b2o, ot, half);
    stream_weights<Cfg::S2 / 4>(wrsrc, wvoff, w2o, [&](int g, const float4 &w) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = 4 * g + u, t = s / Cfg::OT2, ot = s % Cfg::OT2;
        a2[ot] = mfma32(comp(w, u), a1[t >> 4][t & 15], a2[ot]);
      }
    });
#pragma unroll
    for (int ot = 0; ot < Cfg::OT2; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[ot][r] = fmaxf(a2[ot][r], 0.0f);

    // ---- layer 3 (roles flipped): H3 = H2 . W3^T, then max over this tile's 32 points -------------------
    {
      f32x16 a3;
      constexpr int GPT = Cfg::KS2 / 4;  // weight groups per output tile
      stream_weights<Cfg::S3 / 4>(wrsrc, wvoff, w3o, [&](int g, const float4 &w) __attribute__((always_inline)) {
        const int ot = g / GPT, gg = g % GPT;
        if (gg == 0) a3 = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = 4 * gg + u;
          a3 = mfma32(a2[t >> 4][t & 15], comp(w, u), a3);
        }
        if (gg == GPT - 1) {
          float m = a3[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) m = fmaxf(m, a3[r]);
          omax[ot] = fmaxf(omax[ot], m);
        }
      });
    }
  }

  // ---- pooled row: combine the two halves (they hold disjoint points), bias, ReLU, store ----------------
  float *orow = out + qid * out_stride;
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) {
    float v = omax[ot];
    v = mpx_max_across_halves(v);
    const int ch = ot * 32 + col;
    v = fmaxf(v + bias3[ch], 0.0f);
    if (half == 0) orow[ch] = v;
  }
}


// ---- packed variant: only distinct neighbours are evaluated ---------------------------------------------
// Slots [cnt, nsample) of a neighbourhood repeat its first member (ball-query padding).  The MLP is per
// point and max-pooling is idempotent, so evaluating each distinct neighbour once gives a bit-identical
// result.  A wave takes Q consecutive queries, rounds each neighbourhood up to a multiple of GR rows (4 for the
// wide module, 2 for the narrow one; repeating the first neighbour) and packs them back to back into 32-row MFMA
// tiles.  After the last layer a lane holds, per output channel, 16 / GR groups of GR consecutive rows; every group
// belongs to one query, so pooling is a max over each group followed by a merge of the groups in row order that
// flushes the running maximum to the output row whenever the (wave-uniform) query changes.
// One wave per workgroup: waves carry different numbers of tiles, and a multi-wave workgroup would hold
// its SIMD slots until its slowest wave retires.  Large launches are persistent (one workgroup per wave slot of the
// chip, units from a device-side queue -- below).
//
// FACT (factored first layer): W1.[p_j - c_i ; f_j] + b1 = (W1.[p_j ; f_j]) - (W1x.c_i - b1) is linear, so the
// caller evaluates the first term once per POINT (`pre` [B*N, C1], a plain GEMM over the points) and the second
// once per QUERY (`ctr` [B*npoint, C1]); the kernel then starts at relu(pre[j] - ctr[i]) -- a gather, a
// subtraction -- and skips the layer-1 MFMAs of every (query, neighbour) row (15 % of SA2's matrix work).
// Same arithmetic up to the order of that one sum (tolerance-level, not bit-level, vs the direct form).
template <int CF, int C1, int C2, int C3, int Q, bool FACT, int MAXNS = SA_MAX_NSAMPLE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CF == 1 ? 4 : 2, CF == 1 ? 4 : 2)))
    sa_mlp_packed_kernel(const float *__restrict__ xyz, int stride, const float *__restrict__ new_xyz,
                         int new_stride, const float *__restrict__ feat, int feat_stride,
                         const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt, int64_t n_query, int N,
                         int npoint, int nsample, const float *__restrict__ wpack, float *__restrict__ out,
                         int out_stride, int bpe, const float *__restrict__ pre_rows, const float *__restrict__ ctr,
                         int append_centre, unsigned int *__restrict__ queue) {
  using Cfg = SaCfg<CF, C1, C2, C3>;
  // rows of a neighbourhood are rounded up to GR (repeating its first member): 4 for the wide module (62 rows per query on
  // the bench scenes: 2 % of padding, eight pooling groups per tile), 2 for the narrow one (15 rows per query: the padding
  // was 9 % of its matrix work; sixteen pooling groups per tile instead of eight cost less than that)
  constexpr int GR = CF == 1 ? MPX_SA1_GR : 4, NGRP = 32 / GR;
  static_assert(GR == 2 || GR == 4, "pooling groups of 2 or 4 rows");
  __shared__ float ctr_s[FACT ? Q * C1 : 1];
  // LPOOL: a lane's row groups go to their query's row of pool_s by ds_max_f32 (16 rows x 2 output tiles of the narrow
  // module = 2.1 queries per tile on the bench scenes: the register merge below walked 16 groups with a readlane, a compare
  // and a select each and flushed at every boundary -- 3 VALU per MFMA on the counters against 0.8 inside the stream)
  constexpr bool LPOOL = CF == 1 && !FACT && MPX_SA1_LDSPOOL != 0 && MAXNS <= 128;
  __shared__ __attribute__((aligned(16))) float pool_s[LPOOL ? Q * C3 : 1];
  __shared__ __attribute__((aligned(16))) unsigned char qmap_s[Q * MAXNS / GR + 32];  // (row group -> query; rows per query <= nsample <= MAXNS)
  // biases in LDS: they initialise the accumulators at every tile and are added at every flush -- as global loads
  // their latency sits on the critical path of each tile
  __shared__ __attribute__((aligned(16))) float b1_s[C1], b2_s[C2], b3_s[C3];
  static_assert(Q >= 1 && Q <= 32, "queries per wave");
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, col = lane & 31;
  // XCD-aware order: hardware dispatches workgroup h to XCD h % 8 (observed, used for speed only).  All
  // workgroups of one environment are given to one XCD so its cloud / feature rows are fetched into ONE L2
  // instead of eight (bpe = workgroups per environment; falls back to the natural order for ragged grids).
  // PERSISTENT launch (queue != nullptr; one workgroup per wave slot of the chip): the units -- Q consecutive queries --
  // are handed out by a device-side queue, one counter per XCD (an environment's units stay on one XCD, in order), the
  // next one requested when the current one's last tile starts.  Units differ 1 : 30 in rows: as one workgroup per unit the wave
  // slots stood empty 13 % of the kernel (SQ_WAVE_CYCLES: 6.97 of 8 waves per CU resident), and the matrix pipes with them.
  const int64_t n_units = (n_query + Q - 1) / Q;
  const int xcd = blockIdx.x & 7;
  const int64_t j_end = bpe > 0 ? n_units / 8 : n_units;  // units of this workgroup's queue
  int steal = 0;  // queues beyond the own one this wave has moved on to (an XCD that runs dry helps the next one out:
                  // the environments' row counts differ, and so do the sums over an XCD's share of them)
  auto next_unit = [&]() __attribute__((always_inline)) {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(queue + (bpe > 0 ? ((xcd + steal) & 7) : 0), 1u);
    return (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)v);
  };
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wpack), 0, (int)(Cfg::TOTAL * 4), 0x00020000);
  const int wvoff = lane * 16;
  for (int i = threadIdx.x; i < C1; i += 64) b1_s[i] = wpack[Cfg::B1_OFF + i];
  for (int i = threadIdx.x; i < C2; i += 64) b2_s[i] = wpack[Cfg::B2_OFF + i];
  for (int i = threadIdx.x; i < C3; i += 64) b3_s[i] = wpack[Cfg::B3_OFF + i];
  // The weights of the layers this kernel evaluates are ONE contiguous stream of 16-byte-per-lane groups (4 MFMA
  // steps each) in wpack: it is walked through a two-deep register ring that never drains -- the first chunk of a
  // layer is requested while the previous layer still feeds the matrix pipe, and the last chunk of a tile already
  // prefetches the first chunk of the next tile (same addresses; also across units).  Only the very first chunk of a
  // wave is exposed.
  constexpr int G1 = FACT ? 0 : Cfg::S1 / 4, G2 = Cfg::S2 / 4, G3 = Cfg::S3 / 4, GT = G1 + G2 + G3;
  constexpr bool LAZY = FACT && MPX_SA_SPREAD;  // (the factored wide module; C1 / 8 = 16 float4 per gathered row half)
  constexpr int CH = 4, NC0 = (GT + CH - 1) / CH, NC = NC0 + (NC0 & 1);  // even: the ring parity repeats per tile
  constexpr int GPT = Cfg::KS2 / 4;                                      // weight groups per layer-3 output tile
  float4 ring[2][CH];
  auto fetch = [&](int c, int base) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (c * CH + u < GT) ring[c & 1][u] = bload16(wrsrc, wvoff + u * 1024, base + c * CH * 1024);  // (one scalar offset per chunk: the group's 1 KB step rides in the instruction's immediate)
  };
  {
    int wb = (int)(FACT ? Cfg::W2_OFF : Cfg::W1_OFF) * 4;
    asm volatile("" : "+s"(wb));
    fetch(0, wb);
  }
  int64_t j_next = queue ? next_unit() : 0;
  for (bool first = true;; first = false) {
  int64_t wg;
  if (queue) {
    int64_t j = j_next;
    bool dry = false;
    while (j >= j_end) {
      if (bpe == 0 || ++steal == 8) {
        dry = true;
        break;
      }
      j = next_unit();
    }
    if (dry) break;
    wg = bpe > 0 ? ((j / bpe) * 8 + ((xcd + steal) & 7)) * bpe + j % bpe : j;
  } else {
    if (!first) break;
    wg = blockIdx.x;
    if (bpe > 0) {
      const int64_t slot = wg >> 3;
      wg = ((slot / bpe) * 8 + xcd) * bpe + slot % bpe;
    }
  }
  const int64_t q0 = wg * Q;
  const int nq = (int)min((int64_t)Q, n_query - q0);
  if constexpr (LPOOL) {  // (one wave per workgroup: its LDS operations execute in program order -- the previous unit's
                          // read-out below is complete before these stores land)
    const float ninf = -__builtin_inff();
#pragma unroll
    for (int e = lane; e < Q * C3 / 4; e += 64) *reinterpret_cast<float4 *>(pool_s + 4 * e) = make_float4(ninf, ninf, ninf, ninf);
  }
  if constexpr (FACT) {  // this wave's per-query terms -> LDS (read back per row at every tile start)
#pragma unroll
    for (int i = threadIdx.x; i < Q * C1 / 4; i += 64) {
      const int qi = (4 * i) / C1;
      if (qi < nq)
        *reinterpret_cast<float4 *>(ctr_s + 4 * i) = *reinterpret_cast<const float4 *>(ctr + q0 * C1 + 4 * i);
    }
    __syncthreads();
  }

  // lane i < nq: distinct-neighbour count of query q0+i, its row count (multiple of GR) and row offset
  int my_cnt = 0, my_rows = 0;
  if (lane < nq) {
    const int c = cnt[q0 + lane];
    my_cnt = c <= 0 ? 1 : (c > nsample ? nsample : c);  // no hit: the zero-initialised row = point 0
    my_rows = (my_cnt + GR - 1) & ~(GR - 1);
  }
  int pre = my_rows;  // inclusive prefix sum over lanes 0..Q-1
#pragma unroll
  for (int o = 1; o < Q; o <<= 1) {
    const int t = __shfl_up(pre, o);
    if (lane >= o) pre += t;
  }
  const int total = __builtin_amdgcn_readlane(pre, Q - 1);
  // environment of this wave's query `lane` (the 64-bit division happens once, here, not per tile)
  const int my_env = (int)((q0 + (lane < Q ? lane : 0)) / npoint);
  pre -= my_rows;  // exclusive
  // row group -> query of this wave: one byte per GR rows, written by the queries' own lanes; the groups behind the last
  // row (the tail of the last tile) belong to the last query.  (A row's query used to be found by a compare / select
  // chain over the Q row offsets: ~105 VALU instructions per tile for 16 queries, each costing matrix-pipe time.)
  if (lane < nq) {
    const int g1 = (pre + my_rows) / GR;
    for (int g = pre / GR; g < g1; ++g) qmap_s[g] = (unsigned char)lane;
  }
  if (lane < 32 / GR) qmap_s[total / GR + lane] = (unsigned char)(nq - 1);

  __syncthreads();  // (biases / query terms visible; a one-wave workgroup: no more than the LDS wait)
  // register r of a tile holds channel ot*32 + (r&3) + 8*(r>>2) + 4*half
  auto bias_lds = [&](const float *bs, int ot) __attribute__((always_inline)) {
    f32x16 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 q = *reinterpret_cast<const float4 *>(bs + ot * 32 + 8 * g + 4 * half);
      v[4 * g + 0] = q.x;
      v[4 * g + 1] = q.y;
      v[4 * g + 2] = q.z;
      v[4 * g + 3] = q.w;
    }
    return v;
  };

  const int qbase = (int)q0;  // query ids fit 31 bits (checked by the launcher)
  if constexpr (!FACT) {
    // columns [C3, C3+4) of this wave's output rows <- [centre xyz | 0]: the rows become the operand [f | xyz | 0] of
    // the next module's per-point first-layer GEMM (mpx_sa_mlp_factored) without a separate pass over them
    if (append_centre) {
#pragma unroll
      for (int e = lane; e < 4 * Q; e += 64) {  // (16 queries per pass of the wave)
        const int qi = e >> 2, c = e & 3;
        if (qi < nq)
          out[(int64_t)(qbase + qi) * out_stride + C3 + c] = c < 3 ? new_xyz[(int64_t)(qbase + qi) * new_stride + c] : 0.0f;
      }
    }
  }
  float run[Cfg::OT3];  // running max of the query being merged, per output tile (this lane's half of the rows)
  int cur = 0;          // ... and which of this wave's queries that is (wave-uniform; the same for every output tile
                        // between two row tiles, so it advances once per row tile)
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) run[ot] = -__builtin_inff();
  auto flush = [&](int ot, int qi) __attribute__((always_inline)) {
    float v = run[ot];
    v = mpx_max_across_halves(v);
    const int ch = ot * 32 + col;
    v = fmaxf(v + b3_s[ch], 0.0f);
    if (half == 0) out[(int64_t)(qbase + qi) * out_stride + ch] = v;
    run[ot] = -__builtin_inff();
  };
  // row -> (query of this wave, neighbour slot); rows past the end repeat the last query's first slot
  auto map_row = [&](int p, int &qi, int &off) __attribute__((always_inline)) {
    qi = qmap_s[p / GR];
    const int qpre = __builtin_amdgcn_ds_bpermute(4 * qi, pre), qcnt = __builtin_amdgcn_ds_bpermute(4 * qi, my_cnt);
    const int slot = p - qpre;
    off = slot < qcnt ? slot : 0;
  };
  // raw layer-1 inputs of one row: (x - centre) and this lane-half's feature chunk (plain locals: an
  // array inside a struct that is passed around by reference is not promoted to registers)
  constexpr int NF = FACT ? C1 / 2 : (CF == 1 ? 1 : CF / 2);
  float raw_dx, raw_dy, raw_dz, raw_f[NF];
  auto gather = [&](int qi, int k) __attribute__((always_inline)) {
    const int64_t qg = q0 + qi;
    const int64_t b = __shfl(my_env, qi);
    if constexpr (FACT) {  // this lane-half's 4-channel groups of the point's pre-activation row
      const float *pa = pre_rows + (b * N + k) * (int64_t)C1 + 4 * half;
#pragma unroll
      for (int i = 0; i < C1 / 8; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(pa + 8 * i);
        raw_f[4 * i + 0] = v.x;
        raw_f[4 * i + 1] = v.y;
        raw_f[4 * i + 2] = v.z;
        raw_f[4 * i + 3] = v.w;
      }
      return;
    }
    const float *ctr = new_xyz + qg * new_stride;
    const float *pp = xyz + (b * N + k) * (int64_t)stride;
    const float *f = feat + (b * N + k) * (int64_t)feat_stride;
    raw_dx = pp[0] - ctr[0];
    raw_dy = pp[1] - ctr[1];
    raw_dz = pp[2] - ctr[2];
    if (CF == 1) {
      raw_f[0] = f[0];
    } else {
      const float4 *fr = reinterpret_cast<const float4 *>(f + half * (CF / 2));
#pragma unroll
      for (int i = 0; i < CF / 8; ++i) {
        const float4 v = fr[i];
        raw_f[4 * i + 0] = v.x;
        raw_f[4 * i + 1] = v.y;
        raw_f[4 * i + 2] = v.z;
        raw_f[4 * i + 3] = v.w;
      }
    }
  };

  // a quarter of the factored gather (4 of the 16 float4 of this lane-half's row): issued chunk by chunk
  auto gather_part = [&](int qi, int k, int part) __attribute__((always_inline)) {
    if constexpr (FACT) {
      const int64_t b = __shfl(my_env, qi);
      const float *pa = pre_rows + (b * N + k) * (int64_t)C1 + 4 * half;
#pragma unroll
      for (int i = 4 * part; i < 4 * part + 4; ++i) {
        if (i < C1 / 8) {
          const float4 v = *reinterpret_cast<const float4 *>(pa + 8 * i);
          raw_f[4 * i + 0] = v.x;
          raw_f[4 * i + 1] = v.y;
          raw_f[4 * i + 2] = v.z;
          raw_f[4 * i + 3] = v.w;
        }
      }
    }
  };
  // gather pipeline: the neighbour index of the next tile is fetched at tile start, its data during layer 3
  int q_cur, q_next = 0, k_next = 0;
  {
    int off;
    map_row(col, q_cur, off);
    gather(q_cur, idx[(q0 + q_cur) * nsample + off]);
    if (total > 32) {
      map_row(32 + col, q_next, off);
      k_next = idx[(q0 + q_next) * nsample + off];
    }
  }

  for (int rt = 0; rt < total; rt += 32) {
    // the next unit is claimed when the LAST tile of this one starts (the counter's round trip hides behind the tile).
    // Claimed at the start of the unit -- a whole unit ahead -- the units in flight under one L2 spanned twice as many
    // environments (the fetch of this kernel doubled: section 5 of DESIGN.md)
    if (queue && rt + 32 >= total) j_next = next_unit();
    int wb = (int)(FACT ? Cfg::W2_OFF : Cfg::W1_OFF) * 4;
    asm volatile("" : "+s"(wb));

    float x0[Cfg::KS0];
    f32x16 a1[Cfg::OT1];
    if constexpr (FACT) {
      // register r of tile ot = channel 32*ot + 8*(r>>2) + 4*half + (r&3) = group i = 4*ot + (r>>2) of raw_f
      const float *cq = ctr_s + q_cur * C1 + 4 * half;
#pragma unroll
      for (int i = 0; i < C1 / 8; ++i) {
        const float4 c = *reinterpret_cast<const float4 *>(cq + 8 * i);
        a1[i >> 2][4 * (i & 3) + 0] = fmaxf(raw_f[4 * i + 0] - c.x, 0.0f);
        a1[i >> 2][4 * (i & 3) + 1] = fmaxf(raw_f[4 * i + 1] - c.y, 0.0f);
        a1[i >> 2][4 * (i & 3) + 2] = fmaxf(raw_f[4 * i + 2] - c.z, 0.0f);
        a1[i >> 2][4 * (i & 3) + 3] = fmaxf(raw_f[4 * i + 3] - c.w, 0.0f);
      }
    } else {
      x0[0] = half ? raw_dy : raw_dx;
      if (CF == 1) {
        x0[1] = half ? raw_f[0] : raw_dz;
      } else {
        x0[1] = half ? 0.0f : raw_dz;
#pragma unroll
        for (int i = 0; i < CF / 2; ++i) x0[2 + i] = raw_f[i];
      }
    }
    const int q_tile = q_cur;  // which query this lane's row of the current tile belongs to
    // ... per GR-row group, wave-uniform (non-decreasing).  Read once per tile when 8 output tiles share them; the
    // narrow module (2 output tiles, 16 queries' offsets already in SGPRs) reads them where they are used.
    constexpr bool HOIST = Cfg::OT3 > 2;
    int gq_s[HOIST ? 8 : 1];
    if constexpr (HOIST) {
#pragma unroll
      for (int grp = 0; grp < 8; ++grp) gq_s[grp] = __builtin_amdgcn_readlane(q_tile, 4 * grp);  // (HOIST: GR == 4)
    }
    auto gq = [&](int grp) __attribute__((always_inline)) {
      if constexpr (HOIST) return gq_s[grp];
      else return __builtin_amdgcn_readlane(q_tile, GR * grp);
    };
    const int cur0 = cur;
    // LPOOL: the queries of the tile's 16 row groups, one byte each (the same 16 bytes in every lane); whether one query
    // owns the whole tile is wave-uniform
    static_assert(!LPOOL || (GR == 2 && NGRP == 16), "a tile's group -> query bytes are one 16-byte LDS read");
    const int q_gather = q_next, k_gather = k_next;
    q_cur = q_next;
    if (rt + 64 < total) {
      int off;
      map_row(rt + 64 + col, q_next, off);
      k_next = idx[(q0 + q_next) * nsample + off];
    }

    f32x16 a2[Cfg::OT2], a3;
    if constexpr (!FACT) {
#pragma unroll
      for (int ot = 0; ot < Cfg::OT1; ++ot) a1[ot] = bias_lds(b1_s, ot);
    } else {
#pragma unroll
      for (int ot = 0; ot < Cfg::OT2; ++ot) a2[ot] = bias_lds(b2_s, ot);
    }

    // one weight group = 4 MFMA steps of whichever layer it belongs to (g is a compile-time constant after unrolling)
    auto step = [&](int g, const float4 &w) __attribute__((always_inline)) {
      if (g < G1) {  // ---- layer 1: H1^T = W1 . X^T
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int s = 4 * g + u, t = s / Cfg::OT1, ot = s % Cfg::OT1;
          a1[ot] = mfma32(comp(w, u), x0[t], a1[ot]);
        }
        if (g == G1 - 1) {
#pragma unroll
          for (int ot = 0; ot < Cfg::OT1; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[ot][r] = fmaxf(a1[ot][r], 0.0f);
#pragma unroll
          for (int ot = 0; ot < Cfg::OT2; ++ot) a2[ot] = bias_lds(b2_s, ot);
        }
      } else if (g < G1 + G2) {  // ---- layer 2: H2^T = W2 . H1^T (H1 tiles are the B operands in place)
        const int gg = g - G1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int s = 4 * gg + u, t = s / Cfg::OT2, ot = s % Cfg::OT2;
          a2[ot] = mfma32(comp(w, u), a1[t >> 4][t & 15], a2[ot]);
        }
        if (gg == G2 - 1) {
          if constexpr (!LAZY) {
#pragma unroll
            for (int ot = 0; ot < Cfg::OT2; ++ot)
#pragma unroll
              for (int r = 0; r < 16; ++r) a2[ot][r] = fmaxf(a2[ot][r], 0.0f);
            // the next tile's rows are fetched now: layer 3 is long enough to cover the latency and layer 1/2's
            // inputs (the register peak) are gone
            if (rt + 32 < total) gather(q_gather, k_gather);
          }
        }
      } else {
        // ---- layer 3 (roles flipped), pooled per query as each output tile completes.  A lane holds, for its
        // channel, four groups of 4 consecutive rows (group 2j + half = rows 4g..4g+3); groups never straddle
        // queries, so: max inside each group, then merge the 8 groups in row order and flush the running maximum
        // whenever the (wave-uniform) query changes.
        const int g3 = g - G1 - G2, ot = g3 / GPT, gg = g3 % GPT;
        if (gg == 0) a3 = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (LAZY) {
          // LAZY (the wide module): the ReLU of layer 2's result is applied register by register just ahead of its first
          // use (the first output tile's k-steps) and the next tile's rows are requested at the start of the SECOND
          // output tile, four loads per chunk -- neither a 64-instruction VALU block nor 16 loads back to back stand
          // in front of the matrix stream
          if (ot == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int t = 4 * gg + u;
              a2[t >> 4][t & 15] = fmaxf(a2[t >> 4][t & 15], 0.0f);
            }
          }
          if (ot == 1 && gg < 4 && rt + 32 < total) gather_part(q_gather, k_gather, gg);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = 4 * gg + u;
          a3 = mfma32(a2[t >> 4][t & 15], comp(w, u), a3);
        }
        if (gg == GPT - 1) {
          // this lane's group maxima: register 4 j + i holds row 8 j + 4 half + i, so group g (rows GR g ...) lives in the
          // lanes of half (GR g >> 2) & 1 as gm[2 j + (i >> 1)] (GR == 2) or gm[j] (GR == 4)
          float gm[16 / GR];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (GR == 4) {
              gm[j] = fmaxf(fmaxf(a3[4 * j], a3[4 * j + 1]), fmaxf(a3[4 * j + 2], a3[4 * j + 3]));
            } else {
              gm[2 * j] = fmaxf(a3[4 * j], a3[4 * j + 1]);
              gm[2 * j + 1] = fmaxf(a3[4 * j + 2], a3[4 * j + 3]);
            }
          }
          if constexpr (LPOOL) {
            // this lane's 8 groups: gm[2 j + ii] = rows 8 j + 4 half + 2 ii .. + 1 = group 4 j + 2 half + ii of the tile
            // (read where it is used, once per output tile: four registers held across the tile's MFMA stream spill)
            const u32x4 qmv = *reinterpret_cast<const u32x4 *>(qmap_s + rt / GR);
            const bool one_query =
                __builtin_amdgcn_readfirstlane((int)(qmv.x & 0xffu)) == __builtin_amdgcn_readfirstlane((int)(qmv.w >> 24));
            float *prow = pool_s + ot * 32 + col;
            if (one_query) {
              float m = gm[0];
#pragma unroll
              for (int j = 1; j < 8; ++j) m = fmaxf(m, gm[j]);
              (void)__hip_atomic_fetch_max(prow + (qmv.x & 0xffu) * C3, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const unsigned w = (j == 0 ? qmv.x : j == 1 ? qmv.y : j == 2 ? qmv.z : qmv.w) >> (16 * half);
                (void)__hip_atomic_fetch_max(prow + (w & 0xffu) * C3, gm[2 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_max(prow + ((w >> 8) & 0xffu) * C3, gm[2 * j + 1], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          } else if (gq(NGRP - 1) == cur0) {  // the whole tile belongs to the query being merged (the wide module's common case): no flush
            float m = gm[0];
#pragma unroll
            for (int j = 1; j < 16 / GR; ++j) m = fmaxf(m, gm[j]);
            run[ot] = fmaxf(run[ot], m);
          } else {
            int c = cur0;
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
              const int g_q = gq(grp);
              if (g_q != c) {
                flush(ot, c);
                c = g_q;
              }
              // (a select, not a branch)
              const int row0 = GR * grp;  // first row of the group: rows 8 j + 4 h + i
              const float mine = GR == 4 ? gm[row0 >> 3] : gm[2 * (row0 >> 3) + ((row0 & 3) >> 1)];
              run[ot] = fmaxf(run[ot], (((row0 >> 2) & 1) == half) ? mine : -__builtin_inff());
            }
          }
        }
      }
    };
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      fetch((c + 1) % NC, wb);  // the last chunk of a tile requests the first chunk of the next one
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (c * CH + u < GT) step(c * CH + u, ring[c & 1][u]);
      // issue order of the chunk: its weight loads ONE at a time between MFMAs instead of back to back in front of them
      // (a cluster of memory instructions costs matrix-pipe time even with other waves on the SIMD: measured on the
      // group-all kernel, sa3_chain.hip, 0.90 -> 0.98 of the matrix floor inside the layer loops)
      // (the narrow first module, four waves per SIMD, did not gain: 10.8 -> 11.0 ms; it keeps the compiler's order)
      if constexpr (MPX_SA_SPREAD && CF != 1) {
        constexpr int c_ot1 = (G1 + G2 + GPT) / CH;  // first chunk of the second output tile of layer 3
        if (LAZY && c >= c_ot1 && c < c_ot1 + 4) {   // these chunks also carry a quarter of the next tile's gather
#pragma unroll
          for (int u = 0; u < 2 * CH; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one load
          }
        } else {
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one weight load
          }
        }
      }
    }
    if constexpr (!LPOOL) cur = gq(NGRP - 1);
  }
  if constexpr (LPOOL) {
    // the unit's pooled rows: relu(max + b3), 16 bytes per lane (the launcher checked the output rows' alignment)
#pragma unroll 2
    for (int e = lane; e < Q * C3 / 4; e += 64) {
      const int qi = (4 * e) / C3, ch = (4 * e) % C3;
      if (qi < nq) {
        const float4 v = *reinterpret_cast<const float4 *>(pool_s + 4 * e), bb = *reinterpret_cast<const float4 *>(b3_s + ch);
        *reinterpret_cast<float4 *>(out + (int64_t)(qbase + qi) * out_stride + ch) =
            make_float4(fmaxf(v.x + bb.x, 0.0f), fmaxf(v.y + bb.y, 0.0f), fmaxf(v.z + bb.z, 0.0f), fmaxf(v.w + bb.w, 0.0f));
      }
    }
  } else {
#pragma unroll
    for (int ot = 0; ot < Cfg::OT3; ++ot) flush(ot, cur);
  }
  }  // units
}

// ---- host entry points -----------------------------------------------------------------------------------
// wave slots of the current device for a one-wave workgroup at `per_cu` waves per CU (the persistent launches' grid;
// a multiple of 8: the hardware deals workgroups to the XCDs round-robin)
static int64_t sa_wave_slots(int per_cu) {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int n = cus[dev & 63].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
    cus[dev & 63].store(n, std::memory_order_relaxed);
  }
  return ((int64_t)n * per_cu) & ~(int64_t)7;
}
template <int CF, int C1, int C2, int C3>
static int launch_sa(const float *xyz, int stride, const float *new_xyz, int new_stride, const float *feat,
                     int feat_stride, const int32_t *idx, const int32_t *cnt, int B, int N, int npoint, int nsample,
                     const float *wpack, float *out, int out_stride, int append_centre, mpx_stream_t stream) {
  const int64_t nq = (int64_t)B * npoint;
  MPX_REQUIRE(nq / 4 + 1 < ((int64_t)1 << 31), "mpx_sa_mlp: too many query points");
  MPX_REQUIRE(!append_centre || (cnt && out_stride >= C3 + 4),
              "mpx_sa_mlp: append_centre needs the hit counts and out_stride >= c3 + 4");
  MPX_REQUIRE(!append_centre || nsample <= SA_MAX_NSAMPLE, "mpx_sa_mlp: append_centre needs nsample <= %d", SA_MAX_NSAMPLE);
  if (cnt && nsample <= SA_MAX_NSAMPLE) {  // (more slots than the row map holds: every slot is walked -- the same result)
    // queries per wave: 8-16 tiles of work on typical scenes.  A small batch (a single planning problem up to a few
    // dozen) would leave most CUs idle at that size, so it runs QS queries per wave instead: same rows, same
    // arithmetic per row (bit-identical results), 4x the waves and a quarter of the latency.
    constexpr int QL = CF == 1 ? MPX_SA1_Q : 8, QS = CF == 1 ? 4 : 2;
    int rc = 0;
    auto go = [&](auto qtag) {
      constexpr int Q = decltype(qtag)::value;
      int64_t nw = (nq + Q - 1) / Q;
      // whole environments per XCD when the grid is regular (npoint % Q == 0, B % 8 == 0)
      const int bpe = (npoint % Q == 0 && B % 8 == 0) ? npoint / Q : 0;
      unsigned int *queue = nullptr;
      const int64_t slots = sa_wave_slots(CF == 1 ? 16 : 8);
      if (slots > 0 && nw >= 4 * slots) {  // many units per wave slot: one workgroup per slot, units from the device-side queue
        int exhausted = 0;
        queue = mpx_unit_queue_for(mpx_s(stream), &exhausted);
        if (!queue && !exhausted) {
          mpx_set_error("mpx_sa_mlp: cannot reset the unit queue");
          rc = 1;
          return;
        }
        if (queue) nw = slots;  // (no private slot left for this stream: one unit per wave, no queue)
      }
      // (the narrow module with <= 128 slots per neighbourhood and 16-byte aligned output rows: the form with the small
      // row map that pools through LDS, see LPOOL in the kernel)
      bool lpool = false;
      if constexpr (CF == 1 && MPX_SA1_LDSPOOL != 0) {
        lpool = nsample <= 128 && out_stride % 4 == 0 && ((uintptr_t)out & 15) == 0;
        if (lpool)
          hipLaunchKernelGGL((sa_mlp_packed_kernel<CF, C1, C2, C3, Q, false, 128>), dim3((unsigned)nw), dim3(64), 0,
                             mpx_s(stream), xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, nq, N, npoint,
                             nsample, wpack, out, out_stride, bpe, nullptr, nullptr, append_centre, queue);
      }
      if (!lpool)
        hipLaunchKernelGGL((sa_mlp_packed_kernel<CF, C1, C2, C3, Q, false>), dim3((unsigned)nw), dim3(64), 0,
                           mpx_s(stream), xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, nq, N, npoint,
                           nsample, wpack, out, out_stride, bpe, nullptr, nullptr, append_centre, queue);
    };
    if (nq >= 1024 * QL) go(std::integral_constant<int, QL>{});
    else go(std::integral_constant<int, QS>{});
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL((sa_mlp_kernel<CF, C1, C2, C3>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0,
                       mpx_s(stream), xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, nq, N, npoint,
                       nsample, wpack, out, out_stride);
  }
  MPX_LAUNCH_CHECK("mpx_sa_mlp");
}

#define SA_DISPATCH(CALL)                                                          \
  if (C == 1 && c1 == 64 && c2 == 64 && c3 == 64) { CALL(1, 64, 64, 64); }         \
  else if (C == 64 && c1 == 128 && c2 == 128 && c3 == 256) { CALL(64, 128, 128, 256); } \
  else {                                                                           \
    mpx_set_error("mpx_sa: unsupported MLP (C=%d, %d, %d, %d)", C, c1, c2, c3);    \
    return 1;                                                                      \
  }

MPX_EXPORT int mpx_sa_mlp(const float *xyz, int stride, const float *new_xyz, int new_stride,
                          const float *feat, int feat_stride, int C, const int32_t *idx, const int32_t *cnt, int B,
                          int N, int npoint, int nsample, const float *wpack, int c1, int c2, int c3, float *out,
                          int out_stride, int append_centre, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N >= 1 && npoint >= 0, "mpx_sa_mlp: bad size");
  MPX_REQUIRE(nsample > 0 && nsample % 32 == 0, "mpx_sa_mlp: nsample must be a positive multiple of 32");
  MPX_REQUIRE(stride >= 3 && new_stride >= 3 && out_stride >= c3, "mpx_sa_mlp: bad stride");
  MPX_REQUIRE(C == 1 || (feat_stride % 4 == 0 && ((uintptr_t)feat & 15) == 0),
              "mpx_sa_mlp: feature rows must be 16-byte aligned");
  MPX_REQUIRE(((uintptr_t)wpack & 15) == 0, "mpx_sa_mlp: wpack must be 16-byte aligned");
  if (B == 0 || npoint == 0) return 0;
#define CALL(a, b, c, d) \
  return launch_sa<a, b, c, d>(xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, B, N, npoint, nsample, wpack, out, out_stride, append_centre, stream)
  SA_DISPATCH(CALL)
#undef CALL
}

MPX_EXPORT int mpx_sa_mlp_factored(const float *pre, const float *ctr, const int32_t *idx, const int32_t *cnt, int B,
                                   int N, int npoint, int nsample, const float *wpack, int C, int c1, int c2, int c3,
                                   float *out, int out_stride, mpx_stream_t stream) {
  MPX_REQUIRE(C == 64 && c1 == 128 && c2 == 128 && c3 == 256,
              "mpx_sa_mlp_factored: built for the (64+3, 128, 128, 256) module (C=%d, %d, %d, %d)", C, c1, c2, c3);
  MPX_REQUIRE(B >= 0 && N >= 1 && npoint >= 0 && nsample > 0 && nsample <= SA_MAX_NSAMPLE,
              "mpx_sa_mlp_factored: bad size (nsample 1..%d)", SA_MAX_NSAMPLE);
  MPX_REQUIRE(pre && ctr && idx && cnt, "mpx_sa_mlp_factored: NULL operand (hit counts are required)");
  MPX_REQUIRE(out_stride >= c3, "mpx_sa_mlp_factored: bad stride");
  MPX_REQUIRE((((uintptr_t)wpack | (uintptr_t)pre | (uintptr_t)ctr) & 15) == 0,
              "mpx_sa_mlp_factored: operands must be 16-byte aligned");
  if (B == 0 || npoint == 0) return 0;
  const int64_t nq = (int64_t)B * npoint;
  MPX_REQUIRE(nq / 4 + 1 < ((int64_t)1 << 31), "mpx_sa_mlp_factored: too many query points");
  int rc = 0;
  auto go = [&](auto qtag) {  // queries per wave: 8, or 2 / 1 for small batches (see launch_sa)
    constexpr int Q = decltype(qtag)::value;
    int64_t nw = (nq + Q - 1) / Q;
    const int bpe = (npoint % Q == 0 && B % 8 == 0) ? npoint / Q : 0;
    unsigned int *queue = nullptr;
    const int64_t slots = sa_wave_slots(8);
    if (slots > 0 && nw >= 4 * slots) {  // (see launch_sa)
      int exhausted = 0;
      queue = mpx_unit_queue_for(mpx_s(stream), &exhausted);
      if (!queue && !exhausted) {
        mpx_set_error("mpx_sa_mlp_factored: cannot reset the unit queue");
        rc = 1;
        return;
      }
      if (queue) nw = slots;  // (no private slot left for this stream: one unit per wave, no queue)
    }
    hipLaunchKernelGGL((sa_mlp_packed_kernel<64, 128, 128, 256, Q, true>), dim3((unsigned)nw), dim3(64), 0,
                       mpx_s(stream), nullptr, 0, nullptr, 0, nullptr, 0, idx, cnt, nq, N, npoint, nsample, wpack, out,
                       out_stride, bpe, pre, ctr, 0, queue);
  };
  if (nq >= 1024 * MPX_SA2_Q) go(std::integral_constant<int, MPX_SA2_Q>{});
  else if (nq >= 1024) go(std::integral_constant<int, 2>{});
  else go(std::integral_constant<int, 1>{});  // a handful of problems: one query (1-2 tiles) per wave
  if (rc) return rc;
  MPX_LAUNCH_CHECK("mpx_sa_mlp_factored");
}

MPX_EXPORT int64_t mpx_sa_pack_size(int C, int c1, int c2, int c3) {
#define CALL(a, b, c, d) return SaCfg<a, b, c, d>::TOTAL
  if (C == 1 && c1 == 64 && c2 == 64 && c3 == 64) { CALL(1, 64, 64, 64); }
  else if (C == 64 && c1 == 128 && c2 == 128 && c3 == 256) { CALL(64, 128, 128, 256); }
#undef CALL
  return -1;
}

template <int CF, int C1, int C2, int C3>
static int launch_pack(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                       const float *b3, float *wpack, mpx_stream_t stream) {
  using Cfg = SaCfg<CF, C1, C2, C3>;
  hipLaunchKernelGGL(sa_pack_kernel<Cfg>, dim3(cdiv(Cfg::TOTAL, 256)), dim3(256), 0, mpx_s(stream), w1, b1, w2,
                     b2, w3, b3, C1, C2, C3, wpack);
  MPX_LAUNCH_CHECK("mpx_sa_pack_weights");
}

MPX_EXPORT int mpx_sa_pack_weights(const float *w1, const float *b1, const float *w2, const float *b2,
                                   const float *w3, const float *b3, int C, int c1, int c2, int c3,
                                   float *wpack, mpx_stream_t stream) {
#define CALL(a, b, c, d) return launch_pack<a, b, c, d>(w1, b1, w2, b2, w3, b3, wpack, stream)
  SA_DISPATCH(CALL)
#undef CALL
}
