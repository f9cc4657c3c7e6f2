// sa_mlp_bf16.hip -- split-bf16 ("bf16x3") variant of the fused QueryAndGroup + shared MLP + max-pool.
//
// Same role, register-chaining scheme and outputs as sa_mlp.hip (PointnetSAModule.forward body,
// /root/reference/mpinets/model.py:366-382), but every fp32 product x*w is evaluated as
//     x_hi*w_hi + x_hi*w_lo + x_lo*w_hi        (x = x_hi + x_lo, w = w_hi + w_lo, all four bf16)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 MFMAs of 32 cycles per 16 k-values instead
// of 8 fp32 MFMAs of 64 cycles -- 5.3x fewer matrix-pipe cycles.  The dropped x_lo*w_lo term and the
// split residuals are O(2^-16) relative; measured against the fp32 oracle the policy output moves
// by ~3e-7 (tolerance 1e-5), FPS / ball-query indices are untouched (they never see features).
//
// At this rate the weight stream is too wide to be fetched per wave from L2 (85 B/clk/CU), so the
// 8 waves of a workgroup walk the stream in lockstep: chunks of 4 step-tiles (8 KB: a 1 KB w_hi and a
// 1 KB w_lo operand block each) are staged global -> registers -> LDS by all waves together through
// a 2-deep ring, one barrier per chunk, and every wave reads its MFMA operands from LDS
// (16 B/lane ds_read_b128, lane-linear -> conflict free).  L2 traffic per workgroup drops 8x.
// Activations never leave registers: layer outputs (fp32 accumulators, point on the lane axis) are
// split into packed bf16 hi/lo pairs in place and are the next layer's B operand as they are.
#include "common.h"

#include <mutex>
#include <unordered_map>

#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PAD_CH (-1)
#ifndef MPX_BF16_G
#define MPX_BF16_G 8
#endif
#ifndef MPX_BF16_WAVES
#define MPX_BF16_WAVES 8
#endif
constexpr int G = MPX_BF16_G;             // step-tiles per chunk
constexpr int TILE_BYTES = 2048;          // w_hi (64 lanes x 16 B) + w_lo
constexpr int CHUNK_BYTES = G * TILE_BYTES;
constexpr int WAVES = MPX_BF16_WAVES;

template <int CF, int C1, int C2, int C3>
struct BCfg {
  static_assert(CF == 1 || CF % 16 == 0, "feature channels: 1 or a multiple of 16");
  static_assert(C1 % 32 == 0 && C2 % 32 == 0 && C3 % 32 == 0, "layer widths must be multiples of 32");
  static constexpr int CIN = 3 + CF;
  static constexpr int HALF0 = CF == 1 ? 2 : 2 + CF / 2;           // layer-1 inputs held per lane-half
  static constexpr int KS0 = (HALF0 + 7) / 8;                       // K16 steps of layer 1
  static constexpr int KS1 = C1 / 16, KS2 = C2 / 16;                // K16 steps of layers 2, 3
  static constexpr int OT1 = C1 / 32, OT2 = C2 / 32, OT3 = C3 / 32;
  static constexpr int ST1 = KS0 * OT1, ST2 = KS1 * OT2, ST3 = OT3 * KS2;  // real step-tiles per layer
  // every layer is padded to whole chunks (zero-weight dummies) so that layer boundaries -- where
  // accumulators are turned into the next layer's operands -- coincide with chunk boundaries
  static constexpr int ST1P = (ST1 + G - 1) / G * G, ST2P = (ST2 + G - 1) / G * G, ST3P = (ST3 + G - 1) / G * G;
  static constexpr int O2 = ST1P, O3 = ST1P + ST2P;                 // first step-tile of layers 2 and 3
  static constexpr int STP = ST1P + ST2P + ST3P;
  static constexpr int NCH = STP / G;
  static_assert(KS2 % G == 0 || G % KS2 == 0, "layer-3 chunks must align with output tiles");
  static constexpr int OPC = G > KS2 ? G / KS2 : 1;                 // layer-3 output tiles touched by one chunk
  __host__ __device__ static constexpr int layer_of(int st) { return st < O2 ? 1 : (st < O3 ? 2 : 3); }
  __host__ __device__ static constexpr bool is_real(int st) {
    return st < O2 ? st < ST1 : (st < O3 ? st - O2 < ST2 : st - O3 < ST3);
  }
  static constexpr int64_t W_BYTES = (int64_t)STP * TILE_BYTES;
  static constexpr int64_t B1_OFF = W_BYTES, B2_OFF = B1_OFF + 4 * C1, B3_OFF = B2_OFF + 4 * C2;
  static constexpr int64_t TOTAL_BYTES = B3_OFF + 4 * C3;

  // layer-1 input channel held by lane-half h at position i (0..8*KS0-1)
  __host__ __device__ static int chan0(int i, int h) {
    if (CF == 1) return i == 0 ? (h ? 1 : 0) : (i == 1 ? (h ? 3 : 2) : PAD_CH);  // (dx|dy), (dz|label)
    if (i == 0) return h ? 1 : 0;
    if (i == 1) return h ? PAD_CH : 2;
    if (i < 2 + CF / 2) return 3 + h * (CF / 2) + (i - 2);
    return PAD_CH;
  }
  // layers 2/3: K16 step s, element e of lane-half h  ->  channel of the previous layer
  __host__ __device__ static int chan_tile(int s, int e, int h) {
    const int it = s >> 1, r = 8 * (s & 1) + e;
    return it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
  }
};

__device__ __forceinline__ unsigned short f2bf(float x) {  // round-to-nearest-even, like (__bf16)x
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

// ---- weight packing: [step-tile][hi|lo][lane][8 x bf16], then the three fp32 bias vectors ------------------
template <class Cfg>
__global__ void __launch_bounds__(256)
    sa_pack_bf16_kernel(const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
                        const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3,
                        int c1, int c2, int c3, unsigned char *__restrict__ wpack) {
  const int64_t e_id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_w = (int64_t)Cfg::STP * 64 * 8;  // one thread per (step-tile, lane, element): writes hi and lo
  if (e_id < n_w) {
    const int e = (int)(e_id & 7), lane = (int)((e_id >> 3) & 63), st = (int)(e_id >> 9);
    const int h = lane >> 5, o32 = lane & 31;
    float v = 0.0f;
    if (Cfg::is_real(st)) {
      int in, cin, ot;
      const float *w;
      if (st < Cfg::O2) {
        const int s = st / Cfg::OT1;
        ot = st % Cfg::OT1;
        in = Cfg::chan0(8 * s + e, h);
        cin = Cfg::CIN;
        w = w1;
      } else if (st < Cfg::O3) {
        const int q = st - Cfg::O2, s = q / Cfg::OT2;
        ot = q % Cfg::OT2;
        in = Cfg::chan_tile(s, e, h);
        cin = c1;
        w = w2;
      } else {
        const int q = st - Cfg::O3, s = q % Cfg::KS2;
        ot = q / Cfg::KS2;
        in = Cfg::chan_tile(s, e, h);
        cin = c2;
        w = w3;
      }
      if (in != PAD_CH) v = w[(size_t)(ot * 32 + o32) * cin + in];
    }
    const unsigned short hi = f2bf(v);
    const unsigned short lo = f2bf(v - bf2f(hi));
    unsigned short *dst = reinterpret_cast<unsigned short *>(wpack + (int64_t)st * TILE_BYTES);
    dst[lane * 8 + e] = hi;
    dst[512 + lane * 8 + e] = lo;
  }
  const int64_t b_id = e_id - n_w;
  if (b_id >= 0 && b_id < c1 + c2 + c3) {
    float *bd = reinterpret_cast<float *>(wpack + Cfg::B1_OFF);
    bd[b_id] = b_id < c1 ? b1[b_id] : (b_id < c1 + c2 ? b2[b_id - c1] : b3[b_id - c1 - c2]);
  }
}

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}

__device__ __forceinline__ float4 bload16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

__device__ __forceinline__ f32x16 bias_tile(__amdgpu_buffer_rsrc_t rsrc, int bias_off_bytes, int ot, int half) {
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

// relu + split one accumulator tile into the two K16 operand pairs of the next layer
__device__ __forceinline__ void relu_split_tile(const f32x16 &acc, bf16x8 (&hi)[2], bf16x8 (&lo)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(acc[8 * u + e], 0.0f);
    split8(v, hi[u], lo[u]);
  }
}

// The lockstep weight stream of one workgroup.
struct Stream {
  __amdgpu_buffer_rsrc_t rsrc; // the packed weights as a buffer: chunk addresses are scalar offsets
  int voff;                    // this thread's byte offset inside a chunk
  unsigned char *lds;          // ring base
  int lds_slice;               // this lane's byte offset inside a chunk
  int lds_lane;                // lane * 16
  int cur;                     // ring slot holding the chunk being consumed (0/1)
  int next_cc;                 // cyclic index of the chunk to fetch next
  int nch, first;              // the walk covers chunks [first, nch)
  static_assert(CHUNK_BYTES == 2 * 64 * WAVES * 16, "two 16-byte pieces per thread per chunk");
  uint4 stage0, stage1;        // chunk (current + 1), in flight or landed (named members: an array
                               // member is not promoted to registers and lands in scratch)

  __device__ __forceinline__ void fetch() {
    const int soff = next_cc * CHUNK_BYTES;
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + 64 * WAVES * 16, 0);
    stage0 = make_uint4(a.x, a.y, a.z, a.w);
    stage1 = make_uint4(b.x, b.y, b.z, b.w);
    next_cc = next_cc + 1 == nch ? first : next_cc + 1;
  }
  __device__ __forceinline__ void put(int slot) {
    unsigned char *p = lds + slot * CHUNK_BYTES + lds_slice;
    *reinterpret_cast<uint4 *>(p) = stage0;
    *reinterpret_cast<uint4 *>(p + 64 * WAVES * 16) = stage1;
  }
  __device__ __forceinline__ void start() {  // chunk 0 -> slot 0, chunk 1 -> stage
    fetch();
    put(0);
    fetch();
    cur = 0;
    __syncthreads();
  }
  // call after the last operand read of the current chunk
  __device__ __forceinline__ void advance() {
    put(cur ^ 1);
    fetch();
    __syncthreads();
    cur ^= 1;
  }
  __device__ __forceinline__ void operands(int j, bf16x8 &hi, bf16x8 &lo) const {
    const unsigned char *p = lds + cur * CHUNK_BYTES + j * TILE_BYTES + lds_lane;
    hi = *reinterpret_cast<const bf16x8 *>(p);
    lo = *reinterpret_cast<const bf16x8 *>(p + 1024);
  }
};

// raw (fp32) layer-1 inputs of one packed row: neighbour point, its query's centre, its feature half-row
template <int CF>
struct RawIn {
  float px, py, pz, cx, cy, cz;
  float f[CF == 1 ? 1 : CF / 2];
  __device__ __forceinline__ void load(const float *__restrict__ p, const float *__restrict__ c,
                                       const float *__restrict__ fr, int half) {
    px = p[0];
    py = p[1];
    pz = p[2];
    cx = c[0];
    cy = c[1];
    cz = c[2];
    if (CF == 1) {
      f[0] = fr[0];
    } else {
      const float4 *q4 = reinterpret_cast<const float4 *>(fr + half * (CF / 2));
#pragma unroll
      for (int i = 0; i < CF / 8; ++i) {
        const float4 q = q4[i];
        f[4 * i + 0] = q.x;
        f[4 * i + 1] = q.y;
        f[4 * i + 2] = q.z;
        f[4 * i + 3] = q.w;
      }
    }
  }
};

__device__ __forceinline__ f32x16 bias_tile_lds(const float *bias_lds, int ot, int half) {
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = *reinterpret_cast<const float4 *>(bias_lds + ot * 32 + 8 * g + 4 * half);
    v[4 * g + 0] = q.x;
    v[4 * g + 1] = q.y;
    v[4 * g + 2] = q.z;
    v[4 * g + 3] = q.w;
  }
  return v;
}

// Q queries per wave; their DISTINCT neighbours (count from the ball query; padding repeats the first
// neighbour, and max-pooling is idempotent) are rounded up to multiples of 4 rows and packed back to
// back into 32-row tiles.  `order` (optional) lists the queries sorted by row count so that the 8
// lockstep waves of a workgroup carry (nearly) the same number of tiles; the workgroup runs max(tiles).
// (The factored form of the second module -- first layer per point / per query -- is the persistent kernel below.)
template <int CF, int C1, int C2, int C3, int Q>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
    sa_mlp_bf16_kernel(const float *__restrict__ xyz, int stride, const float *__restrict__ new_xyz, int new_stride,
                       const float *__restrict__ feat, int feat_stride, const int32_t *__restrict__ idx,
                       const int32_t *__restrict__ cnt, const int32_t *__restrict__ order, int64_t n_query, int N,
                       int npoint, int nsample, const unsigned char *__restrict__ wpack, float *__restrict__ out,
                       int out_stride) {
  using Cfg = BCfg<CF, C1, C2, C3>;
  constexpr int FIRST = 0;  // first chunk of the walk
  __shared__ __attribute__((aligned(16))) unsigned char ring[2 * CHUNK_BYTES + 4 * (C1 + C2) + 64];
  float *bias_lds = reinterpret_cast<float *>(ring + 2 * CHUNK_BYTES);  // [b1 | b2]
  int *tiles_lds = reinterpret_cast<int *>(ring + 2 * CHUNK_BYTES + 4 * (C1 + C2));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int64_t q0 = ((int64_t)blockIdx.x * WAVES + wave) * Q;
  const bool live = q0 < n_query;  // wave-uniform; dead waves still walk the stream (barriers), store nothing
  const int nq = live ? (int)min((int64_t)Q, n_query - q0) : 0;

  // lane i < nq: global id, distinct-neighbour count, row count (multiple of 4) and row offset of query i
  int my_q = 0, my_cnt = 1, my_rows = 0;
  if (lane < nq) {
    my_q = order ? order[q0 + lane] : (int)(q0 + lane);
    const int c = cnt ? cnt[my_q] : nsample;
    my_cnt = c <= 0 ? 1 : (c > nsample ? nsample : c);  // no hit: the zero-initialised row = point 0
    my_rows = (my_cnt + 3) & ~3;
  }
  int pre = my_rows;
#pragma unroll
  for (int o = 1; o < Q; o <<= 1) {
    const int t = __shfl_up(pre, o);
    if (lane >= o) pre += t;
  }
  const int total = __builtin_amdgcn_readlane(pre, Q - 1);
  pre -= my_rows;
  int s_pre[Q], s_cnt[Q], s_q[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    s_pre[i] = __builtin_amdgcn_readlane(pre, i);
    s_cnt[i] = __builtin_amdgcn_readlane(my_cnt, i);
    s_q[i] = __builtin_amdgcn_readlane(my_q, i);
  }
  if (lane == 0) tiles_lds[wave] = total <= 32 ? 32 : ((total + 31) & ~31);

  for (int i = threadIdx.x; i < C1 + C2; i += 64 * WAVES)
    bias_lds[i] = reinterpret_cast<const float *>(wpack + Cfg::B1_OFF)[i];
  Stream ws;
  ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(wpack), 0, (int)Cfg::TOTAL_BYTES, 0x00020000);
  ws.voff = threadIdx.x * 16;
  ws.lds = ring;
  ws.lds_slice = threadIdx.x * 16;
  ws.lds_lane = lane * 16;
  ws.first = FIRST;
  ws.next_cc = FIRST;
  ws.nch = Cfg::NCH;
  ws.start();  // (its barrier also publishes the biases and the per-wave row counts)
  int n_rows = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) n_rows = max(n_rows, tiles_lds[w]);
  const float *bias3 = reinterpret_cast<const float *>(wpack + Cfg::B3_OFF);

  // row -> (global query id, neighbour index); rows past this wave's end repeat its last query's slot 0
  auto map_row = [&](int p, int &qg, int &nb_off) {
    int qpre = 0, qcnt = s_cnt[0];
    qg = s_q[0];
#pragma unroll
    for (int i = 1; i < Q; ++i) {
      const bool ge = i < nq && p >= s_pre[i];
      qg = ge ? s_q[i] : qg;
      qpre = ge ? s_pre[i] : qpre;
      qcnt = ge ? s_cnt[i] : qcnt;
    }
    const int slot = p - qpre;
    nb_off = slot < qcnt ? slot : 0;
  };
  auto gather = [&](RawIn<CF> &raw, int qg, int k) {
    const int64_t b = qg / npoint;
    raw.load(xyz + (b * N + k) * (int64_t)stride, new_xyz + (int64_t)qg * new_stride,
             feat + (b * N + k) * (int64_t)feat_stride, half);
  };

  float run[Cfg::OT3];  // running max of the query being merged, per output tile (this lane's half of the rows)
  int cur[Cfg::OT3];    // ... and which query that is (wave-uniform)
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) {
    run[ot] = -__builtin_inff();
    cur[ot] = s_q[0];
  }
  auto flush = [&](int ot, int qg) {
    float v = run[ot];
    v = mpx_max_across_halves(v);
    const int ch = ot * 32 + col;
    v = fmaxf(v + bias3[ch], 0.0f);
    if (live && half == 0) out[(int64_t)qg * out_stride + ch] = v;
    run[ot] = -__builtin_inff();
  };

  // gather pipeline: neighbour index one tile ahead (issued at tile start), neighbour data issued in layer 3
  RawIn<CF> raw;
  int q_cur, q_next = 0, k_next = 0;
  {
    int off;
    map_row(col, q_cur, off);
    const int k0 = idx[(int64_t)q_cur * nsample + off];
    gather(raw, q_cur, k0);
    if (n_rows > 32) {
      map_row(32 + col, q_next, off);
      k_next = idx[(int64_t)q_next * nsample + off];
    }
  }

  for (int rt = 0; rt < n_rows; rt += 32) {
    // ---- layer-1 operands from the prefetched row, split hi/lo -------------------------------------------------
    bf16x8 xh[Cfg::KS0], xl[Cfg::KS0];
    f32x16 a1[Cfg::OT1], a2[Cfg::OT2];
    bf16x8 h1[Cfg::OT1][2], l1[Cfg::OT1][2];
    {
      float v[8 * Cfg::KS0];
#pragma unroll
      for (int i = 0; i < 8 * Cfg::KS0; ++i) v[i] = 0.0f;
      const float dx = raw.px - raw.cx, dy = raw.py - raw.cy, dz = raw.pz - raw.cz;
      v[0] = half ? dy : dx;
      if (CF == 1) {
        v[1] = half ? raw.f[0] : dz;
      } else {
        v[1] = half ? 0.0f : dz;
#pragma unroll
        for (int i = 0; i < CF / 2; ++i) v[2 + i] = raw.f[i];
      }
#pragma unroll
      for (int s = 0; s < Cfg::KS0; ++s) {
        float t8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t8[e] = v[8 * s + e];
        split8(t8, xh[s], xl[s]);
      }
    }
    const int q_tile = q_cur;  // query of this lane's row in the tile being computed
    const int q_gather = q_next, k_gather = k_next;
    q_cur = q_next;
    if (rt + 64 < n_rows) {  // index of the row after next: issued now, consumed a tile later
      int off;
      map_row(rt + 64 + col, q_next, off);
      k_next = idx[(int64_t)q_next * nsample + off];
    }
    // The next tile's row data is fetched inside the chunk walk below, at the first chunk of layer 3
    // (register pressure peaks in layer 2; layer 3 is long enough to cover the latency).
    constexpr int GATHER_CHUNK = Cfg::O3 / G;

    bf16x8 h2[Cfg::OT2][2], l2[Cfg::OT2][2];
    f32x16 a3[Cfg::OPC][2];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT1; ++ot) a1[ot] = bias_tile_lds(bias_lds, ot, half);

#pragma unroll
    for (int c = FIRST; c < Cfg::NCH; ++c) {
      if (c == GATHER_CHUNK && rt + 32 < n_rows) gather(raw, q_gather, k_gather);
      // ---- operands that become available / are first needed in this chunk ---------------------------------
      if (c * G == Cfg::O2) {
#pragma unroll
        for (int ot = 0; ot < Cfg::OT2; ++ot) a2[ot] = bias_tile_lds(bias_lds + C1, ot, half);
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int st = c * G + j;
        if (!Cfg::is_real(st)) continue;
        if (st >= Cfg::O2 && st < Cfg::O3) {  // layer 2, first use of input tile s>>1 (ot == 0, even s)
          const int q = st - Cfg::O2, s = q / Cfg::OT2, ot = q % Cfg::OT2;
          if (ot == 0 && (s & 1) == 0) relu_split_tile(a1[s >> 1], h1[s >> 1], l1[s >> 1]);
        } else if (st >= Cfg::O3) {            // layer 3, first output tile walks the input tiles in order
          const int q = st - Cfg::O3, s = q % Cfg::KS2, ot = q / Cfg::KS2;
          if (ot == 0 && (s & 1) == 0) relu_split_tile(a2[s >> 1], h2[s >> 1], l2[s >> 1]);
          if (s == 0) {
            a3[ot % Cfg::OPC][0] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            a3[ot % Cfg::OPC][1] = a3[ot % Cfg::OPC][0];
          }
        }
      }
      // operands are fetched SUB step-tiles at a time (register budget); within a sub-group three passes
      // (hi*hi, lo*hi, hi*lo) so that consecutive MFMAs hit different accumulators
      constexpr int SUB = 4;
#pragma unroll
      for (int g0 = 0; g0 < G; g0 += SUB) {
        bf16x8 wh[SUB], wl[SUB];
#pragma unroll
        for (int j = 0; j < SUB; ++j)
          if (Cfg::is_real(c * G + g0 + j)) ws.operands(g0 + j, wh[j], wl[j]);
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
          for (int j = 0; j < SUB; ++j) {
            const int st = c * G + g0 + j;
            if (!Cfg::is_real(st)) continue;
            const bf16x8 w = pass == 1 ? wl[j] : wh[j];
            if (st < Cfg::O2) {
              const int s = st / Cfg::OT1, ot = st % Cfg::OT1;
              a1[ot] = mfma_bf16(w, pass == 2 ? xl[s] : xh[s], a1[ot]);
            } else if (st < Cfg::O3) {
              const int q = st - Cfg::O2, s = q / Cfg::OT2, ot = q % Cfg::OT2;
              a2[ot] = mfma_bf16(w, pass == 2 ? l1[s >> 1][s & 1] : h1[s >> 1][s & 1], a2[ot]);
            } else {
              const int q = st - Cfg::O3, s = q % Cfg::KS2, ot = q / Cfg::KS2;
              // roles flipped: activations are the A operand, weights the B operand
              a3[ot % Cfg::OPC][s & 1] =
                  mfma_bf16(pass == 2 ? l2[s >> 1][s & 1] : h2[s >> 1][s & 1], w, a3[ot % Cfg::OPC][s & 1]);
            }
          }
        }
      }
      ws.advance();
      // ---- end of an output tile of layer 3: pool its rows per query -----------------------------------------
      // a lane holds, for its channel, four groups of 4 consecutive rows (group g = 2j + half = rows 4g..4g+3);
      // groups never straddle queries, so: max inside each group, then merge the 8 groups in row order and
      // flush the running maximum whenever the (wave-uniform) query changes.
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int st = c * G + j;
        if (st < Cfg::O3 || !Cfg::is_real(st)) continue;
        const int q = st - Cfg::O3;
        if (q % Cfg::KS2 == Cfg::KS2 - 1) {
          const int ot = q / Cfg::KS2;
          const f32x16 &u0 = a3[ot % Cfg::OPC][0], &u1 = a3[ot % Cfg::OPC][1];
          float gm[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            gm[jj] = fmaxf(fmaxf(u0[4 * jj] + u1[4 * jj], u0[4 * jj + 1] + u1[4 * jj + 1]),
                           fmaxf(u0[4 * jj + 2] + u1[4 * jj + 2], u0[4 * jj + 3] + u1[4 * jj + 3]));
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const int gq = __builtin_amdgcn_readlane(q_tile, 4 * g);
            if (gq != cur[ot]) {
              flush(ot, cur[ot]);
              cur[ot] = gq;
            }
            if ((g & 1) == half) run[ot] = fmaxf(run[ot], gm[g >> 1]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) flush(ot, cur[ot]);
}

// ---- weight-resident form for a module whose whole pack fits LDS (the first module: 1+3 -> 64 -> 64 -> 64, 18 real
// step-tiles = 36 KB) -------------------------------------------------------------------------------------------------
// The lockstep kernel above walks an 8-wave workgroup through the weight stream with one barrier per chunk; for this
// module the stream is 36 KB and the walk costs more than the matrix work (54 MFMAs = 1.7 k pipe cycles per 32-row
// tile).  Here a workgroup copies the real step-tiles and the biases into LDS once and its four waves are independent
// from then on (no barrier, no ring, no sorting pass): a wave packs the distinct neighbours of Q consecutive queries into
// 32-row tiles exactly as above, reads every weight operand from LDS (lane-linear 16-byte reads), and keeps the same
// arithmetic (split products, fp32 accumulate, exact fp32 bias) and operand layouts -- the pack of mpx_sa_pack_bf16x3 is
// used as it is.  Four waves per SIMD cover each other's gather / split / pooling phases.
#ifndef MPX_RES_OCC
#define MPX_RES_OCC 4
#endif
namespace res {
constexpr int WV = 4;
}
template <int CF, int C1, int C2, int C3, int Q, bool ONE_ENV>
__global__ void __launch_bounds__(64 * res::WV) __attribute__((amdgpu_waves_per_eu(MPX_RES_OCC, MPX_RES_OCC)))
    sa_mlp_bf16_resident_kernel(const float *__restrict__ xyz, int stride, const float *__restrict__ new_xyz,
                                int new_stride, const float *__restrict__ feat, int feat_stride,
                                const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt, int64_t n_query, int N,
                                int npoint, int nsample, const unsigned char *__restrict__ wpack, float *__restrict__ out,
                                int out_stride, int wpe, int row16, int append_centre) {
  using Cfg = BCfg<CF, C1, C2, C3>;
  static_assert(CF == 1, "written for the one-feature first module (layer-1 operand forming)");
  constexpr int NT = Cfg::ST1 + Cfg::ST2 + Cfg::ST3;  // real step-tiles
  constexpr int T2 = Cfg::ST1, T3 = Cfg::ST1 + Cfg::ST2;  // first LDS tile of layers 2, 3
  constexpr int ROWQ = (Q * 128 + 64) / 4 + 16;  // 4-row groups of a wave's rows (nsample <= 128) + the prefetch overhang
  __shared__ __attribute__((aligned(16))) unsigned char wlds[NT * TILE_BYTES + 4 * (C1 + C2 + C3) + res::WV * ROWQ];
  static_assert(sizeof(wlds) <= 40960, "four workgroups per CU");
  unsigned char *rowq_all = wlds + NT * TILE_BYTES + 4 * (C1 + C2 + C3);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, col = lane & 31;
  {  // real step-tiles (the pack pads every layer to whole chunks: skip the dummies) and b1 | b2 | b3
    const uint4 *src = reinterpret_cast<const uint4 *>(wpack);
    uint4 *dst = reinterpret_cast<uint4 *>(wlds);
    constexpr int PER = TILE_BYTES / 16;
    for (int i = threadIdx.x; i < NT * PER; i += 64 * res::WV) {
      const int t = i / PER, r = i % PER;
      const int st = t < T2 ? t : (t < T3 ? Cfg::O2 + (t - T2) : Cfg::O3 + (t - T3));
      dst[i] = src[st * PER + r];
    }
    float *b = reinterpret_cast<float *>(wlds + NT * TILE_BYTES);
    for (int i = threadIdx.x; i < C1 + C2 + C3; i += 64 * res::WV)
      b[i] = reinterpret_cast<const float *>(wpack + Cfg::B1_OFF)[i];
  }
  __syncthreads();  // the only barrier
  const float *b1_s = reinterpret_cast<const float *>(wlds + NT * TILE_BYTES), *b2_s = b1_s + C1, *b3_s = b2_s + C2;
  // (the LDS image never changes, so the compiler would hoist every operand read out of the tile loop -- 36 KB of
  // "loop invariants" per wave, spilled to scratch: the lane's base offset is made opaque once per tile instead)
  int w_off = lane * 16;
  auto operands = [&](int t, bf16x8 &hi, bf16x8 &lo) __attribute__((always_inline)) {
    hi = *reinterpret_cast<const bf16x8 *>(wlds + w_off + t * TILE_BYTES);
    lo = *reinterpret_cast<const bf16x8 *>(wlds + w_off + t * TILE_BYTES + 1024);
  };

  // XCD-aware order (hardware dispatches workgroup h to XCD h % 8): all workgroups of an environment on one XCD, so its
  // cloud is fetched into one L2 (wpe = workgroups per environment; 0: natural order).  (A persistent form with a
  // device-side unit queue, as in the second module's kernel below, was measured 7 % slower here: the SIMDs' issue
  // ports, not their wave slots, are what is full.)
  int64_t wg = blockIdx.x;
  if (wpe > 0) {
    const int64_t xcd = wg & 7, slot = wg >> 3;
    wg = ((slot / wpe) * 8 + xcd) * wpe + slot % wpe;
  }
  const int64_t q0 = (wg * res::WV + wave) * Q;
  if (q0 >= n_query) return;  // (after the barrier)
  const int nq = (int)min((int64_t)Q, n_query - q0);

  // lane i < nq: distinct-neighbour count, row count (multiple of 4) and row offset of query q0 + i
  int my_cnt = 1, my_rows = 0;
  if (lane < nq) {
    const int c = cnt ? cnt[q0 + lane] : nsample;
    my_cnt = c <= 0 ? 1 : (c > nsample ? nsample : c);  // no hit: the zero-initialised row = point 0
    my_rows = (my_cnt + 3) & ~3;
  }
  int pre = my_rows;
#pragma unroll
  for (int o = 1; o < Q; o <<= 1) {
    const int t = __shfl_up(pre, o);
    if (lane >= o) pre += t;
  }
  const int total = __builtin_amdgcn_readlane(pre, Q - 1);
  pre -= my_rows;
  const int n_rows = total <= 32 ? 32 : ((total + 31) & ~31);
  // group of 4 rows -> local query, one byte per group in this wave's LDS strip (a query's rows are whole groups; the
  // groups past the end -- the index prefetch runs two tiles ahead -- belong to the last query): each query's lane
  // writes its own groups, so a tile's row -> query map is one LDS read instead of a compare chain over the Q queries
  unsigned char *rowq = rowq_all + wave * ROWQ;
  {
    const int g0 = pre >> 2, ng = my_rows >> 2;
    for (int j = 0; j < ng; ++j) rowq[g0 + j] = (unsigned char)lane;
    const int gt = total >> 2, ge = (n_rows + 64) >> 2;  // ge - gt <= 24
    if (gt + lane < ge) rowq[gt + lane] = (unsigned char)(nq - 1);
  }
  __builtin_amdgcn_wave_barrier();
  // uniform bases: everything a row needs is a 32-bit offset from one of these (ONE_ENV: npoint is a multiple of Q, the
  // wave's queries share their environment's cloud; the rows are then fetched with buffer loads: one address VGPR)
  const int64_t b0 = q0 / npoint;
  const float *ctr_w = new_xyz + q0 * new_stride;
  float *out_w = out + q0 * out_stride;
  const __amdgpu_buffer_rsrc_t r_idx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(idx + q0 * nsample), 0, Q * nsample * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_ctr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ctr_w), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_xyz = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(xyz + b0 * N * (int64_t)stride), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_feat = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(feat + b0 * N * (int64_t)feat_stride), 0, 0x7ffffff0, 0x00020000);
  auto load_idx = [&](int ql, int off) __attribute__((always_inline)) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r_idx, (ql * nsample + off) * 4, 0, 0);
  };
  if (append_centre && lane < 4 * nq) {  // [centre xyz | 0] behind the pooled features: the next module's operand row
    const int qi = lane >> 2, c = lane & 3;
    out_w[qi * out_stride + C3 + c] = c < 3 ? ctr_w[qi * new_stride + c] : 0.0f;
  }

  // row -> (local query, neighbour slot); rows past the end repeat the last query's first slot
  auto map_row = [&](int p, int &ql, int &off) __attribute__((always_inline)) {
    const int n = rowq[p >> 2];
    ql = n;
    const int qpre = __builtin_amdgcn_ds_bpermute(4 * n, pre), qcnt = __builtin_amdgcn_ds_bpermute(4 * n, my_cnt);
    const int slot = p - qpre;
    off = slot < qcnt ? slot : 0;
  };
  float px, py, pz, pf, cx, cy, cz;  // the gathered row: neighbour point, its label, its query's centre
  auto gather = [&](int ql, int k) __attribute__((always_inline)) {
    if constexpr (ONE_ENV) {
      if (row16) {  // (x, y, z, label) is one aligned 16-byte row of the caller's slab
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_xyz, k * 16, 0, 0);
        px = __uint_as_float(v.x), py = __uint_as_float(v.y), pz = __uint_as_float(v.z), pf = __uint_as_float(v.w);
      } else {
        const int o = k * stride * 4;
        px = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_xyz, o, 0, 0));
        py = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_xyz, o, 4, 0));
        pz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_xyz, o, 8, 0));
        pf = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_feat, k * feat_stride * 4, 0, 0));
      }
    } else {
      const int64_t b = (q0 + ql) / npoint;
      const float *pp = xyz + (b * N + k) * (int64_t)stride;
      px = pp[0], py = pp[1], pz = pp[2];
      pf = feat[(b * N + k) * (int64_t)feat_stride];
    }
    const int o = ql * new_stride * 4;
    cx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_ctr, o, 0, 0));
    cy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_ctr, o, 4, 0));
    cz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_ctr, o, 8, 0));
  };

  float run[Cfg::OT3], b3v[Cfg::OT3];
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) {
    run[ot] = -__builtin_inff();
    b3v[ot] = b3_s[ot * 32 + col];
  }
  int cur = 0;  // local query being merged (wave-uniform)
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out_w, 0, 0x7ffffff0, 0x00020000);
  auto flush = [&](int ql) __attribute__((always_inline)) {  // every output tile of query ql (scalar row offset)
#pragma unroll
    for (int ot = 0; ot < Cfg::OT3; ++ot) {
      const float v = fmaxf(mpx_max_across_halves(run[ot]) + b3v[ot], 0.0f);
      if (half == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_out, (ot * 32 + col) * 4, ql * out_stride * 4, 0);
      run[ot] = -__builtin_inff();
    }
  };

  // row pipeline: neighbour index two tiles ahead, row data one tile ahead
  int ql_cur, ql_next = 0, k_next = 0;
  {
    int off;
    map_row(col, ql_cur, off);
    gather(ql_cur, load_idx(ql_cur, off));
    map_row(32 + col, ql_next, off);
    k_next = load_idx(ql_next, off);
  }
  int half_t = half;
  for (int rt = 0; rt < n_rows; rt += 32) {
    asm volatile("" : "+v"(w_off), "+v"(half_t));
    // ---- layer-1 operand of this lane's row: (dx | dy), (dz | label) by lane-half, zero beyond -------------------
    bf16x8 xh, xl;
    {
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      v[0] = half ? py - cy : px - cx;
      v[1] = half ? pf : pz - cz;
      split8(v, xh, xl);
    }
    const int ql_tile = ql_cur, ql_gather = ql_next, k_gather = k_next;
    ql_cur = ql_next;
    {
      int off;
      map_row(rt + 64 + col, ql_next, off);
      k_next = load_idx(ql_next, off);
    }
    // ---- layer 1: H1t = W1 . Xt ------------------------------------------------------------------------------------
    f32x16 a1[Cfg::OT1];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT1; ++ot) a1[ot] = bias_tile_lds(b1_s, ot, half_t);
    static_assert(Cfg::KS0 == 1, "one K16 step in layer 1");
#pragma unroll
    for (int ot = 0; ot < Cfg::OT1; ++ot) {
      bf16x8 wh, wl;
      operands(ot, wh, wl);
      a1[ot] = mfma_bf16(wh, xh, a1[ot]);
      a1[ot] = mfma_bf16(wl, xh, a1[ot]);
      a1[ot] = mfma_bf16(wh, xl, a1[ot]);
    }
    gather(ql_gather, k_gather);  // the next tile's row (consumed at the top of the next iteration)
    bf16x8 h1[Cfg::OT1][2], l1[Cfg::OT1][2];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT1; ++ot) relu_split_tile(a1[ot], h1[ot], l1[ot]);
    // ---- layer 2: H2t = W2 . H1t, the output tiles side by side (consecutive MFMAs on different accumulators) -------
    f32x16 a2[Cfg::OT2];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT2; ++ot) a2[ot] = bias_tile_lds(b2_s, ot, half_t);
#pragma unroll
    for (int s = 0; s < Cfg::KS1; ++s) {
      bf16x8 wh[Cfg::OT2], wl[Cfg::OT2];
#pragma unroll
      for (int ot = 0; ot < Cfg::OT2; ++ot) operands(T2 + s * Cfg::OT2 + ot, wh[ot], wl[ot]);
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int ot = 0; ot < Cfg::OT2; ++ot)
          a2[ot] = mfma_bf16(pass == 1 ? wl[ot] : wh[ot], pass == 2 ? l1[s >> 1][s & 1] : h1[s >> 1][s & 1], a2[ot]);
    }
    bf16x8 h2[Cfg::OT2][2], l2[Cfg::OT2][2];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT2; ++ot) relu_split_tile(a2[ot], h2[ot], l2[ot]);
    // ---- layer 3 (roles flipped: activations are A, weights B; rows on the register axis, channels on the lanes) ----
    f32x16 a3[Cfg::OT3];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT3; ++ot) a3[ot] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < Cfg::KS2; ++s) {
      bf16x8 wh[Cfg::OT3], wl[Cfg::OT3];
#pragma unroll
      for (int ot = 0; ot < Cfg::OT3; ++ot) operands(T3 + ot * Cfg::KS2 + s, wh[ot], wl[ot]);
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int ot = 0; ot < Cfg::OT3; ++ot)
          a3[ot] = mfma_bf16(pass == 2 ? l2[s >> 1][s & 1] : h2[s >> 1][s & 1], pass == 1 ? wl[ot] : wh[ot], a3[ot]);
    }
    // ---- pooling: a lane holds, for its channel, four groups of 4 consecutive rows (group g = 2j + half = rows
    // 4g .. 4g+3); groups never straddle queries: max inside each group, merge the 8 groups in row order, flush the
    // running maximum whenever the (wave-uniform) query changes
    int gq[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) gq[g] = __builtin_amdgcn_readlane(ql_tile, 4 * g);
    float gm[Cfg::OT3][4];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT3; ++ot)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        gm[ot][jj] = fmaxf(fmaxf(a3[ot][4 * jj], a3[ot][4 * jj + 1]), fmaxf(a3[ot][4 * jj + 2], a3[ot][4 * jj + 3]));
    if (gq[7] == cur) {  // the whole tile belongs to the query being merged
#pragma unroll
      for (int ot = 0; ot < Cfg::OT3; ++ot)
        run[ot] = fmaxf(run[ot], fmaxf(fmaxf(gm[ot][0], gm[ot][1]), fmaxf(gm[ot][2], gm[ot][3])));
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (gq[g] != cur) {
          flush(cur);
          cur = gq[g];
        }
#pragma unroll
        for (int ot = 0; ot < Cfg::OT3; ++ot)
          run[ot] = fmaxf(run[ot], ((g & 1) == half) ? gm[ot][g >> 1] : -__builtin_inff());
      }
    }
  }
  flush(cur);
}

// ---- v2 of the factored (64+3, 128, 128, 256) module: persistent workgroups, ONE software-pipelined wave per SIMD ----
// The lockstep kernel above keeps its matrix pipes ~40 % busy: its non-matrix work (forming relu(pre - ctr) and
// splitting it into bf16 hi / lo, splitting the layer-2 accumulators, pooling, operand reads -- ~1000-1400 instructions
// per 288 MFMAs) sits in phases of its own, and a v_mfma_f32_32x32x16_bf16 occupies the pipe for only 32 cycles, so two
// waves per SIMD cannot cover each other's phases (one wave per SIMD runs the same code at the same speed: measured).
// This variant is built the way MI355X_MICROARCH.md describes a 512-register wave: ONE wave per SIMD that interleaves
// <= 4 single-issue instructions behind every MFMA, by construction:
//   * one persistent workgroup of 4 waves per CU (grid = CU count); a wave takes units of Q = 8 consecutive queries in
//     the order of a device-side queue (one counter per XCD: all queries of an environment on one XCD, its 512 x 128
//     first-layer rows stay in one L2), so no sorting pass is needed and no wave ever waits for another (one barrier,
//     after the LDS fill);
//   * the layer-3 weights (64 step-tiles x (1 KB hi + 1 KB lo) = 128 KB) live in LDS for the whole kernel, the layer-2
//     weights (64 KB per tile) stream from L2 through a 4-stage register ring, three K16 steps ahead;
//   * every layer runs two output tiles at a time on two accumulators (consecutive MFMAs never depend on each other);
//   * the tile loop is a software pipeline written out in issue order (sched_barrier fences keep it): pair A of layer
//     2 | pair B + the split of pair A's accumulators | output pair 0 of layer 3 + the split of pair B | output pairs
//     1-3 + the pooling of the pair before + the NEXT tile's relu(pre - ctr) + split, whose rows were requested when
//     layer 2 ended.
// Arithmetic, operand layouts and the weight pack are exactly those of sa_mlp_bf16_kernel<64,128,128,256,Q,true>.
namespace v2 {
using Cfg = BCfg<64, 128, 128, 256>;
constexpr int Q = 8, WV = 4;
constexpr int C1 = 128, C2 = 128, C3 = 256;
constexpr int W2_OFF = Cfg::O2 * TILE_BYTES;                 // first layer-2 step-tile (s-major, output tile inner)
constexpr int W3_OFF = Cfg::O3 * TILE_BYTES;                 // first layer-3 step-tile (output tile major, s inner)
constexpr int W3_BYTES = Cfg::ST3 * TILE_BYTES;              // 131072
constexpr int LDS_BIAS2 = W3_BYTES, LDS_BIAS3 = LDS_BIAS2 + 4 * C2, LDS_CTR = LDS_BIAS3 + 4 * C3;
constexpr int MAX_NSAMPLE = 128;                             // the launcher refuses larger neighbourhoods: ROWQ below is sized for this
constexpr int ROWQ = (Q * MAX_NSAMPLE + 64) / 4 + 16;        // 4-row groups of a unit's rows (nsample <= MAX_NSAMPLE) + the index prefetch's overhang
constexpr int LDS_ROWQ = LDS_CTR + WV * Q * C1 * 4;          // row group -> local query, one byte each, a strip per wave
constexpr int LDS_BYTES = LDS_ROWQ + WV * ROWQ;              // 150144
constexpr int RS = 8, RD = 7;  // layer-2 ring: stages, prefetch distance in K16 steps (7 x 192 pipe cycles of L2 latency cover)
static_assert(Cfg::ST2 == 32 && Cfg::ST3 == 64 && Cfg::KS1 == 8 && Cfg::KS2 == 8 && Cfg::OT2 == 4 && Cfg::OT3 == 8, "shape");
}  // namespace v2

#define V2_FENCE() __builtin_amdgcn_sched_barrier(0)

// Unit queues of the persistent kernel: 8 counters (one per XCD) per launch, in device memory that belongs to the
// library image (nothing is allocated).  ONE SLOT PER (device, stream): the launches of a stream are ordered (memset,
// kernel, memset, kernel, ...), so they can share a slot, and launches on different streams never alias -- two engines
// that share a model on two streams each get their own counters.  The slot is zeroed on the
// launch stream in front of the kernel (stream-ordered, hipGraph-capturable).  Limits, stated in include/mpinets_hip.h:
// 256 distinct (device, stream) handles per process get a slot; a later handle gets NONE (`*exhausted` = 1, nullptr):
// slots are never shared between streams -- two persistent kernels on one set of counters would each skip the units
// the other claimed and leave output rows unwritten -- so the fp32 launchers fall back to their one-unit-per-wave
// grids and the bf16x3 launcher reports an error.  A captured graph bakes its capture stream's slot in, so two graphs
// captured on the SAME stream must not be replayed concurrently on different streams.
__device__ unsigned int sa2_unit_queues[256 * 8];
unsigned int *mpx_unit_queue_for(hipStream_t stream, int *exhausted) {
  static std::mutex mu;
  static std::unordered_map<unsigned long long, int> slot_of;  // (device << 56) ^ stream handle -> slot
  static unsigned int *base[64];
  if (exhausted) *exhausted = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  unsigned int *q;
  {
    std::lock_guard<std::mutex> lock(mu);
    unsigned int *&b = base[dev & 63];
    if (!b) {
      void *p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(sa2_unit_queues)) != hipSuccess) return nullptr;
      b = static_cast<unsigned int *>(p);
    }
    const unsigned long long key = ((unsigned long long)(dev & 63) << 56) ^ (unsigned long long)(uintptr_t)stream;
    auto it = slot_of.find(key);
    if (it == slot_of.end()) {
      if (slot_of.size() >= (size_t)mpx_unit_queue_slots()) {
        if (exhausted) *exhausted = 1;
        return nullptr;
      }
      it = slot_of.emplace(key, (int)slot_of.size()).first;
    }
    q = b + 8 * it->second;
  }
  if (hipMemsetAsync(q, 0, 8 * sizeof(unsigned int), stream) != hipSuccess) return nullptr;
  return q;
}
// (verification hook: tests shrink the slot count to reach the exhausted path without creating 256 streams)
static std::atomic<int> unit_queue_slots{256};
int mpx_unit_queue_slots() { return unit_queue_slots.load(); }
void mpx_unit_queue_set_slots(int n) { unit_queue_slots.store(n < 0 ? 0 : n > 256 ? 256 : n); }

template <bool PROBE = false>
__global__ void __launch_bounds__(64 * v2::WV) __attribute__((amdgpu_waves_per_eu(1, 1)))
    sa2_bf16x3_persistent_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt, int64_t n_query, int N,
                                 int npoint, int nsample, const unsigned char *__restrict__ wpack, float *__restrict__ out,
                                 int out_stride, const float *__restrict__ pre_rows, const float *__restrict__ ctr,
                                 int xcd_aware, unsigned int *__restrict__ queue, long long *__restrict__ probe) {
  // PROBE (measurement only, tools/probes/sa2_bf16_phase_probe.py): s_memtime stamps of workgroup 100, wave 0 at the
  // start of a tile, after its layer 2 and after its layer 3, for the first 20 tiles it runs
  int pi = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (PROBE) {
      if (blockIdx.x == 100 && threadIdx.x == 0 && pi < 80) probe[pi] = (long long)__builtin_amdgcn_s_memtime();
      ++pi;
    }
  };
  using namespace v2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, col = lane & 31;
  {  // layer-3 weights + the two bias vectors -> LDS, once
    const uint4 *src = reinterpret_cast<const uint4 *>(wpack + W3_OFF);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < W3_BYTES / 16; i += 64 * WV) dst[i] = src[i];
    float *b2 = reinterpret_cast<float *>(smem + LDS_BIAS2);
    for (int i = threadIdx.x; i < C2 + C3; i += 64 * WV)
      b2[i] = reinterpret_cast<const float *>(wpack + Cfg::B2_OFF)[i];  // b2 | b3 are contiguous in the pack
  }
  __syncthreads();  // the only barrier: from here on the waves are independent
  const float *bias2_s = reinterpret_cast<const float *>(smem + LDS_BIAS2);
  const float *bias3_s = reinterpret_cast<const float *>(smem + LDS_BIAS3);
  float b3v[Cfg::OT3];  // this lane's channel of every layer-3 output tile
#pragma unroll
  for (int ot = 0; ot < Cfg::OT3; ++ot) b3v[ot] = bias3_s[ot * 32 + col];
  float *ctr_w = reinterpret_cast<float *>(smem + LDS_CTR) + wave * Q * C1;
  unsigned char *rowq = smem + LDS_ROWQ + wave * ROWQ;
  const unsigned char *w3_lane = smem + lane * 16;
  int w3_off_hi = lane * 16 + 65536;
  asm volatile("" : "+v"(w3_off_hi));  // (opaque: otherwise the compiler folds it back into w3_lane + a 17-bit constant)

  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(wpack), 0, (int)Cfg::TOTAL_BYTES, 0x00020000);
  const int wvoff = lane * 16;
  // layer-2 operand ring: one stage = the (hi, lo) blocks of TWO output tiles (a pair) at one K16 step; running
  // step n = pair * 8 + s lives in stage n % RS
  u32x4 ring[RS][4];
  auto fetch2 = [&](int n) __attribute__((always_inline)) {
    const int pair = n >> 3, s = n & 7;
    // one scalar offset per stage; the four 1 KB blocks (hi, lo of the pair's two tiles) sit in the instruction's
    // 12-bit immediate (lane * 16 + 3072 < 4096): a fourth of the s_mov's of one-offset-per-load
    const int base = W2_OFF + (s * 4 + 2 * pair) * TILE_BYTES;
#pragma unroll
    for (int k = 0; k < 4; ++k) ring[n % RS][k] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + 1024 * k, base, 0);
  };
  // ... one block of a stage (round 3): the four loads of a stage go out ONE per MFMA gap, not back to back -- a cluster
  // of memory instructions in front of a lone wave's MFMAs costs matrix-pipe time (sa3_chain.hip: 0.90 -> 0.98 of the floor)
  auto fetch2_part = [&](int n, int k) __attribute__((always_inline)) {
    const int pair = n >> 3, s = n & 7;
    const int base = W2_OFF + (s * 4 + 2 * pair) * TILE_BYTES;
    ring[n % RS][k] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + 1024 * k, base, 0);
  };
  auto as_bf = [](const u32x4 &v) __attribute__((always_inline)) { return __builtin_bit_cast(bf16x8, v); };
#pragma unroll
  for (int n = 0; n < RD; ++n) fetch2(n);

  // (Round 4, s_memtime probe -- tools/probes/sa2_bf16_phase_probe.py: the compiler sinks most of the next tile's
  // relu(pre - ctr) + split behind the tile's last MFMA, 1.7 k cycles between two tiles.  Pinning every quantum in its gap
  // with an empty volatile asm on its operand registers moves those cycles INTO the layer loops and leaves the tile time
  // where it was (14.9 k / 17.9 k cycles with / without a query boundary): the fillers are not what the tile waits for.)
  // relu + hi / lo split of elements (2e, 2e+1) of a 16-float accumulator tile into the packed bf16 operand pairs = one
  // "quantum" (~8 VALU; relu_split_tile, two elements at a time).
  // Half quanta (round 4): a gap between two MFMAs of a lone in-order wave hides about three plain VALU; a fourth costs
  // ~2 cycles, a burst of 9-13 (a whole quantum) stalls the next MFMA for the burst's own issue time (micro-benchmark:
  // tools/probes/mfma_bf16_stream.hip).  So a quantum is cut in two -- relu + hi, then lo -- and the halves go into
  // consecutive gaps; the empty volatile asm pins each half where it is written (the compiler otherwise sinks most of
  // them behind the tile's last MFMA).
  float tv0 = 0.0f, tv1 = 0.0f;  // relu'd pair between the two halves of a split
  auto split_h = [&](const f32x16 &acc, bf16x8 (&hi)[2], bf16x8 (&lo)[2], int e, int part) __attribute__((always_inline)) {
    const int u = e >> 2, k = 2 * (e & 3);
    if (part == 0) {
      tv0 = fmaxf(acc[8 * u + k], 0.0f), tv1 = fmaxf(acc[8 * u + k + 1], 0.0f);
      hi[u][k] = (__bf16)tv0;
      hi[u][k + 1] = (__bf16)tv1;
      asm volatile("" : "+v"(hi[u]), "+v"(tv0), "+v"(tv1));
    } else {
      lo[u][k] = (__bf16)(tv0 - (float)hi[u][k]);
      lo[u][k + 1] = (__bf16)(tv1 - (float)hi[u][k + 1]);
      asm volatile("" : "+v"(lo[u]));
    }
  };

  // ---- the units of this wave: handed out by a device-side queue (one counter per XCD: an environment's units go to
  // the waves of one XCD, in order); the next unit is requested when the current one's last tile starts.  Unit sizes differ 1 : 30:
  // with a static stride the waves were resident 87 % of the kernel. ---------------------------------------------
  const int64_t n_units = (n_query + Q - 1) / Q;
  const int upe = npoint / Q;  // units per environment (xcd_aware only)
  const int xcd = xcd_aware ? (blockIdx.x & 7) : 0;
  const int64_t j_end = xcd_aware ? (n_query / npoint / 8) * upe : n_units;  // units of this queue
  int steal = 0;  // queues beyond the own one this wave has moved on to (an XCD that runs dry helps the next one out:
                  // the sums of the row counts over an XCD's share of the environments differ by a few per cent)
  auto next_unit = [&]() __attribute__((always_inline)) {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(queue + ((xcd + steal) & 7), 1u);
    return (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)v);
  };
  int64_t j_next = next_unit();
  while (true) {
    int64_t j = j_next;
    bool dry = false;
    while (j >= j_end) {
      if (!xcd_aware || ++steal == 8) {
        dry = true;
        break;
      }
      j = next_unit();
    }
    if (dry) break;
    const int64_t unit = xcd_aware ? ((j / upe) * 8 + ((xcd + steal) & 7)) * upe + j % upe : j;
    const int64_t q0 = unit * Q;
    const int nq = (int)min((int64_t)Q, n_query - q0);
    int my_cnt = 1, my_rows = 0, my_env = 0;
    if (lane < nq) {
      const int c = cnt[q0 + lane];
      my_cnt = c <= 0 ? 1 : (c > nsample ? nsample : c);  // no hit: the zero-initialised row = point 0
      my_rows = (my_cnt + 3) & ~3;
      my_env = (int)((q0 + lane) / npoint);
    }
    int pre = my_rows;
#pragma unroll
    for (int o = 1; o < Q; o <<= 1) {
      const int t = __shfl_up(pre, o);
      if (lane >= o) pre += t;
    }
    const int total = __builtin_amdgcn_readlane(pre, Q - 1);
    pre -= my_rows;
    const int n_rows = total <= 32 ? 32 : ((total + 31) & ~31);
    // group of 4 rows -> local query, one byte per group in this wave's LDS strip (a query's rows are whole groups; the
    // groups past the end -- the index prefetch runs two tiles ahead -- belong to the last query): each query's lane
    // writes its own groups, so a tile's row -> query map is one LDS read + three ds_bpermute instead of a compare /
    // select chain over the Q queries in front of every tile (round 5; the chain also held 3 Q scalar registers)
    {
      const int g0 = pre >> 2, ng = my_rows >> 2;
      for (int j = 0; j < ng; ++j) rowq[g0 + j] = (unsigned char)lane;
      const int gt = total >> 2, ge = (n_rows + 64) >> 2;  // ge - gt <= 24
      if (gt + lane < ge) rowq[gt + lane] = (unsigned char)(nq - 1);
    }
    // this unit's per-query first-layer terms -> this wave's LDS rows (Q x 32 float4: Q / 2 per lane)
#pragma unroll
    for (int u = 0; u < Q / 2; ++u) {
      const int i = lane + 64 * u, qi = i >> 5, c4 = i & 31;
      if (qi < nq)
        *reinterpret_cast<float4 *>(ctr_w + qi * C1 + 4 * c4) =
            *reinterpret_cast<const float4 *>(ctr + (q0 + qi) * C1 + 4 * c4);
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t *const idx_unit = idx + q0 * nsample;  // (uniform: a row's neighbour index is a 32-bit offset from here)

    // row -> (local query, environment, neighbour slot); rows past the end repeat the last query's first slot.  In three
    // steps so that the tile loop can spread them over three gaps: table read | the query's lane values | the slot
    auto map_q = [&](int p) __attribute__((always_inline)) { return (int)rowq[p >> 2]; };
    auto map_fetch = [&](int n, int &qpre, int &qcnt, int &env) __attribute__((always_inline)) {
      qpre = __builtin_amdgcn_ds_bpermute(4 * n, pre);
      qcnt = __builtin_amdgcn_ds_bpermute(4 * n, my_cnt);
      env = __builtin_amdgcn_ds_bpermute(4 * n, my_env);
    };
    auto map_row = [&](int p, int &qi, int &env, int &off) __attribute__((always_inline)) {
      int qpre, qcnt;
      qi = map_q(p);
      map_fetch(qi, qpre, qcnt, env);
      const int slot = p - qpre;
      off = slot < qcnt ? slot : 0;
    };
    float raw_pre[C1 / 2];  // this lane-half's 4-channel groups of the gathered first-layer row
    auto gather = [&](int env, int k) __attribute__((always_inline)) {
      const float *pa = pre_rows + ((int64_t)env * N + k) * (int64_t)C1 + 4 * half;
#pragma unroll
      for (int i = 0; i < C1 / 8; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(pa + 8 * i);
        raw_pre[4 * i + 0] = v.x, raw_pre[4 * i + 1] = v.y, raw_pre[4 * i + 2] = v.z, raw_pre[4 * i + 3] = v.w;
      }
    };
    auto gather_part = [&](int env, int k, int i) __attribute__((always_inline)) {  // one of the row's 16 float4
      const float *pa = pre_rows + ((int64_t)env * N + k) * (int64_t)C1 + 4 * half;
      const float4 v = *reinterpret_cast<const float4 *>(pa + 8 * i);
      raw_pre[4 * i + 0] = v.x, raw_pre[4 * i + 1] = v.y, raw_pre[4 * i + 2] = v.z, raw_pre[4 * i + 3] = v.w;
    };
    float run[Cfg::OT3];
#pragma unroll
    for (int ot = 0; ot < Cfg::OT3; ++ot) run[ot] = -__builtin_inff();
    int cur = 0;  // local query being merged (wave-uniform)
    // (round 5) A tile with exactly ONE query boundary -- every other tile -- pools both sides in one straight block per
    // output tile: the lane's four group maxima are clipped against +-inf limits (limo[j] = +inf where group 2j + half
    // belongs to the query that ends in this tile), the finished query's maximum is completed and stored (both lane
    // halves store the same value: no exec mask), the rest starts the new query's running maximum.  Round 4's form -- up
    // to eight flushes per output tile behind scalar branches -- cost such a tile ~3 k of its ~16 k cycles; it still
    // serves tiles with two or more boundaries.
    // (the lane's eight layer-3 biases sit in registers and the unit's output rows start at one 64-bit base: a flush in
    // the middle of a tile -- a query boundary, every other tile -- used to wait for an LDS read and to rebuild a 64-bit
    // row address with quarter-rate integer multiplies, 3.3 k of a 12.7 k-cycle layer 3; s_memtime probe, round 4)
    float *const out_unit = out + q0 * out_stride + col;
    auto flush = [&](int ot, int qi) __attribute__((always_inline)) {
      float v = run[ot];
      v = mpx_max_across_halves(v);
      v = fmaxf(v + b3v[ot], 0.0f);
      if (half == 0) out_unit[__builtin_amdgcn_readfirstlane(qi) * out_stride + ot * 32] = v;
      run[ot] = -__builtin_inff();
    };

    // layer-2 operands of the tile being computed; formed for the NEXT tile during layer 3 of this one
    bf16x8 h1[4][2], l1[4][2];
    float4 cq[4];  // the four ctr float4 of the 32-channel group being formed
    // quantum g (0..31) of relu(pre - ctr) + split for the row in raw_pre and its query's LDS row `cqp`: channels
    // 4-group i = g >> 1 (tile ot = i >> 2, register group i & 3), element pair (g & 1)
    auto form_q = [&](const float *cqp, int g) __attribute__((always_inline)) {
      const int i = g >> 1, ot = i >> 2, gg = i & 3, pr2 = g & 1;
      if (pr2 == 0) cq[gg] = *reinterpret_cast<const float4 *>(cqp + 8 * i);
      const float c0 = pr2 ? cq[gg].z : cq[gg].x, c1 = pr2 ? cq[gg].w : cq[gg].y;
      const float v0 = fmaxf(raw_pre[4 * i + 2 * pr2] - c0, 0.0f), v1 = fmaxf(raw_pre[4 * i + 2 * pr2 + 1] - c1, 0.0f);
      // accumulator register r = 4 * gg + 2 * pr2 (+1) of tile ot -> operand pair u = r >> 3, element r & 7
      const int r = 4 * gg + 2 * pr2, u = r >> 3, k = r & 7;
      const __bf16 h0 = (__bf16)v0, hh1 = (__bf16)v1;
      h1[ot][u][k] = h0;
      h1[ot][u][k + 1] = hh1;
      l1[ot][u][k] = (__bf16)(v0 - (float)h0);
      l1[ot][u][k + 1] = (__bf16)(v1 - (float)hh1);
    };

    float fv0 = 0.0f, fv1 = 0.0f;
    auto load_cq = [&](const float *cqp, int i) __attribute__((always_inline)) {  // the four ctr values of channel group i
      cq[i & 3] = *reinterpret_cast<const float4 *>(cqp + 8 * i);
    };
    auto form_h = [&](const float *cqp, int g, int part) __attribute__((always_inline)) {
      const int i = g >> 1, ot = i >> 2, gg = i & 3, pr2 = g & 1;
      const int r = 4 * gg + 2 * pr2, u = r >> 3, k = r & 7;
      if (part == 0) {
        const float c0 = pr2 ? cq[gg].z : cq[gg].x, c1 = pr2 ? cq[gg].w : cq[gg].y;
        fv0 = fmaxf(raw_pre[4 * i + 2 * pr2] - c0, 0.0f), fv1 = fmaxf(raw_pre[4 * i + 2 * pr2 + 1] - c1, 0.0f);
        h1[ot][u][k] = (__bf16)fv0;
        h1[ot][u][k + 1] = (__bf16)fv1;
        asm volatile("" : "+v"(h1[ot][u]), "+v"(fv0), "+v"(fv1));
      } else {
        if (pr2 == 1 && i + 1 < C1 / 8) load_cq(cqp, i + 1);  // the next group's ctr values: read a gap ahead of their use
        l1[ot][u][k] = (__bf16)(fv0 - (float)h1[ot][u][k]);
        l1[ot][u][k + 1] = (__bf16)(fv1 - (float)h1[ot][u][k + 1]);
        asm volatile("" : "+v"(l1[ot][u]));
      }
    };

    // ---- per-tile pooling state and pieces (declared once per unit: the tile loop's gap fillers refer to them) ----------
    int ql_tile = 0;
    int gq[8];
    unsigned m_cur = 0, m_last = 0;
    int g1 = 8, cur_off = 0;
    bool one_b = false;
    float limo[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // +inf where this lane's group 2j + half lies in front of the boundary, else -inf
    // boundary shape of the tile (wave-uniform), from two ballots over the rows (row r = lane r): g1 = leading 4-row
    // groups that still belong to `cur`; one_b = every other row belongs to the tile's last query (the local query
    // index never decreases along the rows, so that is ONE boundary)
    auto shape_a = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 8; ++g) gq[g] = __builtin_amdgcn_readlane(ql_tile, 4 * g);
      m_cur = (unsigned)__builtin_amdgcn_ballot_w64(ql_tile == cur);
    };
    auto shape_b = [&]() __attribute__((always_inline)) {
      m_last = (unsigned)__builtin_amdgcn_ballot_w64(ql_tile == gq[7]);
      g1 = (m_cur == 0xffffffffu ? 32 : __builtin_ctz(~m_cur)) >> 2;
      one_b = gq[7] != cur && (m_cur | m_last) == 0xffffffffu;
      const int lim = g1 - half;  // group 2j + half is in front of the boundary <=> 2j < g1 - half
#pragma unroll
      for (int j = 0; j < 4; ++j) limo[j] = __uint_as_float((((unsigned)(lim - (2 * j + 1))) & 0x80000000u) | 0x7f800000u);
      cur_off = cur * out_stride;  // (scalar: the finished query's output row, relative to the unit's first)
      if constexpr (PROBE) {  // boundary shape of the tile beside its stamps: 0 none, 1 one boundary, 2 more
        if (blockIdx.x == 100 && threadIdx.x == 0 && (pi >> 2) < 24) probe[96 + (pi >> 2)] = g1 == 8 ? 0 : (one_b ? 1 : 2);
      }
    };
    f32x16 a3[2][2];
    auto pool = [&](int pr, int part) __attribute__((always_inline)) {  // output pair pr, tile o = part
      const int ot = 2 * pr + part;
      const f32x16 &a = a3[pr & 1][part];
      float gm[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        gm[jj] = fmaxf(fmaxf(a[4 * jj], a[4 * jj + 1]), fmaxf(a[4 * jj + 2], a[4 * jj + 3]));
      if (gq[7] == cur) {  // the whole tile belongs to the query being merged (the common case)
        run[ot] = fmaxf(run[ot], fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3])));
      } else {
        int c = cur;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (gq[g] != c) {
            flush(ot, c);
            c = gq[g];
          }
          run[ot] = fmaxf(run[ot], ((g & 1) == half) ? gm[g >> 1] : -__builtin_inff());
        }
      }
    };
    float gmb[2][4];  // group maxima of the output pair being pooled (pool_gm / pool_fin: pool() in five pieces)
    auto pool_gm = [&](int pr, int part, int jj) __attribute__((always_inline)) {
      const f32x16 &a = a3[pr & 1][part];
      gmb[part][jj] = fmaxf(fmaxf(a[4 * jj], a[4 * jj + 1]), fmaxf(a[4 * jj + 2], a[4 * jj + 3]));
      asm volatile("" : "+v"(gmb[part][jj]));
    };
    auto pool_fin = [&](int pr, int part) __attribute__((always_inline)) {
      const int ot = 2 * pr + part;
      const float *gm = gmb[part];
      if (gq[7] == cur) {  // the whole tile belongs to the query being merged (the common case)
        run[ot] = fmaxf(run[ot], fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3])));
      } else if (one_b) {  // one boundary: the finished query's side is completed and stored, the rest starts the new one
        const float o0 = fminf(gm[0], limo[0]), o1 = fminf(gm[1], limo[1]), o2 = fminf(gm[2], limo[2]), o3 = fminf(gm[3], limo[3]);
        const float n0 = fminf(gm[0], -limo[0]), n1 = fminf(gm[1], -limo[1]), n2 = fminf(gm[2], -limo[2]), n3 = fminf(gm[3], -limo[3]);
        const float done = mpx_max_across_halves(fmaxf(fmaxf(run[ot], fmaxf(o0, o1)), fmaxf(o2, o3)));
        out_unit[cur_off + ot * 32] = fmaxf(done + b3v[ot], 0.0f);
        run[ot] = fmaxf(fmaxf(n0, n1), fmaxf(n2, n3));
      } else {
        int c = cur;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (gq[g] != c) {
            flush(ot, c);
            c = gq[g];
          }
          run[ot] = fmaxf(run[ot], ((g & 1) == half) ? gm[g >> 1] : -__builtin_inff());
        }
      }
    };
    // row pipeline: the neighbour index of a tile's row is loaded a tile before its first-layer row is gathered (at the
    // end of layer 2 of the tile before), which is formed during layer 3 of that tile
    int ql_cur, ql_next = 0, env_next = 0, k_next = 0;
    {  // unit prologue: the first tile's rows, formed without cover (once per ~16 tiles)
      int env, off;
      map_row(col, ql_cur, env, off);
      const int k0 = idx_unit[ql_cur * nsample + off];
      gather(env, k0);
      map_row(32 + col, ql_next, env_next, off);
      k_next = idx_unit[ql_next * nsample + off];
      const float *cqp = ctr_w + ql_cur * C1 + 4 * half;
#pragma unroll
      for (int g = 0; g < 32; ++g) form_q(cqp, g);
    }

    for (int rt = 0; rt < n_rows; rt += 32) {
      // the next unit is claimed when the LAST tile of this one starts (the counter's round trip hides behind the tile;
      // claimed a whole unit ahead, the units in flight under one L2 span twice as many environments -- the fp32
      // kernel's fetch halved with this line, sa_mlp.hip)
      if (rt + 32 >= n_rows) j_next = next_unit();
      stamp();
      ql_tile = ql_cur;
      const int env_gather = env_next, k_gather = k_next;
      const float *cq_next = ctr_w + ql_next * C1 + 4 * half;  // LDS row of the next tile's query (this lane's row)
      ql_cur = ql_next;
      V2_FENCE();
      int mr_n = 0, mr_pre = 0, mr_cnt = 0, mr_env = 0;  // (GAPMAP: the map of row rt + 64 + col between its three steps)
      V2_FENCE();
      // ---- layer 2: Ht = W . Xt, pair A then pair B; MFMA m of a pair = (s, pass, o) = (m / 6, (m % 6) / 2, m % 2) -----
      f32x16 a2[4];
      bf16x8 h2[4][2], l2[4][2];
      a2[0] = bias_tile_lds(bias2_s, 0, half);
      a2[1] = bias_tile_lds(bias2_s, 1, half);
      V2_FENCE();
#pragma unroll
      for (int m = 0; m < 96; ++m) {
        const int pair = m / 48, mm = m % 48, s = mm / 6, pass = (mm % 6) / 2, o = mm % 2, n = pair * 8 + s;
        if (mm % 6 < 4 && n + RD < 16) fetch2_part(n + RD, mm % 6);  // (the ring never holds more than RS = RD + 1 steps)
        a2[2 * pair + o] = mfma_bf16(as_bf(ring[n % RS][2 * o + (pass == 1 ? 1 : 0)]),
                                     pass == 2 ? l1[s >> 1][s & 1] : h1[s >> 1][s & 1], a2[2 * pair + o]);
        // pair A's accumulators are final after MFMA 47: their relu + split rides behind pair B's MFMAs, one HALF quantum
        // per gap in two gaps of three; the third kind of gap (no weight fetch: mm % 6 >= 4) carries one load of the
        // next tile's rows, consumed from output pair 1 of layer 3 on
        if (pair == 1 && mm % 3 != 0) {
          const int hh = (mm / 3) * 2 + (mm % 3 - 1), q = hh >> 1;
          split_h(a2[q >> 3], h2[q >> 3], l2[q >> 3], q & 7, hh & 1);
        }
        if (pair == 1 && mm % 6 >= 4) gather_part(env_gather, k_gather, (mm / 6) * 2 + (mm % 6 - 4));
        // pair A's fetch-free gaps (mm % 6 >= 4; nothing else rides there): the map of the row after next in three
        // steps an LDS round trip apart, the tile's boundary shape (needed from layer 3's second output pair on), pair
        // B's biases
        if (pair == 0) {
          if (mm == 4) mr_n = map_q(rt + 64 + col);
          if (mm == 10) map_fetch(mr_n, mr_pre, mr_cnt, mr_env);
          if (mm == 5) shape_a();
          if (mm == 11) shape_b();
          if (mm == 22) {
            const int slot = rt + 64 + col - mr_pre;
            ql_next = mr_n, env_next = mr_env;
            k_next = idx_unit[mr_n * nsample + (slot < mr_cnt ? slot : 0)];
          }
          if (mm == 28) a2[2] = bias_tile_lds(bias2_s, 2, half);
          if (mm == 34) a2[3] = bias_tile_lds(bias2_s, 3, half);
        }
        V2_FENCE();
      }
      stamp();
      // the next tile's rows are requested now: layer 3 reads LDS only, so these slower loads are not in front of
      // anything the matrix stream waits for; they are consumed from output pair 2 on (~3000 cycles from here)
      // (round 3: one gather load per MFMA gap behind the first 16 MFMAs of layer 3, not 16 back to back here)
      V2_FENCE();
      // ---- layer 3 (roles flipped: activations are A, weights B), output tiles in pairs, weights from LDS ------------
      bf16x8 w3r[3][4];  // operand stages: running step n3 = pr * 8 + s lives in stage n3 % 3, read two steps ahead
      // (a ds_read's immediate offset is 16 bits: the upper 64 KB of the layer-3 weights are read from a second base
      // register -- one address per kernel instead of a v_add_u32 in front of every other read, 64 per tile)
      auto w3_at = [&](int off) __attribute__((always_inline)) {
        return off < 65536 ? w3_lane + off : smem + w3_off_hi + (off - 65536);
      };
      auto load3 = [&](int n3) __attribute__((always_inline)) {
        const int pr = n3 >> 3, s = n3 & 7;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const unsigned char *p = w3_at(((2 * pr + o) * 8 + s) * TILE_BYTES);
          w3r[n3 % 3][2 * o] = *reinterpret_cast<const bf16x8 *>(p);
          w3r[n3 % 3][2 * o + 1] = *reinterpret_cast<const bf16x8 *>(p + 1024);
        }
      };
      auto load3_part = [&](int n3, int j) __attribute__((always_inline)) {  // one of the step's four LDS reads
        const int pr = n3 >> 3, s = n3 & 7, o = j >> 1;
        const unsigned char *p = w3_at(((2 * pr + o) * 8 + s) * TILE_BYTES + (j & 1) * 1024);
        w3r[n3 % 3][j] = *reinterpret_cast<const bf16x8 *>(p);
      };
      load3(0);
      load3(1);
      V2_FENCE();
#pragma unroll
      for (int m = 0; m < 192; ++m) {
        const int pr = m / 48, mm = m % 48, s = mm / 6, pass = (mm % 6) / 2, o = mm % 2, n3 = pr * 8 + s;
        if (mm % 6 < 4 && n3 + 2 < 32) load3_part(n3 + 2, mm % 6);
        {
          const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          const bf16x8 x = pass == 2 ? l2[s >> 1][s & 1] : h2[s >> 1][s & 1];
          const bf16x8 w = w3r[n3 % 3][2 * o + (pass == 1 ? 1 : 0)];
          a3[pr & 1][o] = mfma_bf16(x, w, (s == 0 && pass == 0) ? zero : a3[pr & 1][o]);
        }
        // fillers, one small piece per gap:
        if (pr == 0) {  // pair B's split in half quanta: tile 2 in gaps 0..15 (needed from s = 4), tile 3 in gaps 16..31
          if (mm < 32) split_h(a2[2 + mm / 16], h2[2 + mm / 16], l2[2 + mm / 16], (mm % 16) >> 1, mm & 1);
          if (mm == 40) load_cq(cq_next, 0);
        } else {
          // pooling of the pair before, in the gaps with mm % 3 == 0: the four group maxima of a tile one per gap, then
          // the merge into the running maximum (or the flushes of a tile with a query boundary)
          if (mm % 3 == 0 && mm < 30) {
            const int step = mm / 3, part = step / 5, piece = step % 5;
            if (piece < 4) pool_gm(pr - 1, part, piece);
            else pool_fin(pr - 1, part);
          }
          // the NEXT tile's layer-2 operands in half quanta: 32 halves in output pair 1 (two gaps of three), 16 each in
          // pairs 2 and 3 (one gap of three)
          if (pr == 1 && mm % 3 != 0) {
            const int hh = (mm / 3) * 2 + (mm % 3 - 1);
            form_h(cq_next, hh >> 1, hh & 1);
          }
          if (pr >= 2 && mm % 3 == 1) {
            const int hh = 32 + (pr - 2) * 16 + mm / 3;
            form_h(cq_next, hh >> 1, hh & 1);
          }
        }
        // the first layer-2 stages of the NEXT tile (the weights never change): one block per gap over the last MFMAs
        if (m >= 192 - 4 * RD) fetch2_part((m - (192 - 4 * RD)) >> 2, (m - (192 - 4 * RD)) & 3);
        V2_FENCE();
      }
#pragma unroll
      for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) pool_gm(3, part, jj);
        pool_fin(3, part);
      }
      (void)pool;
      cur = gq[7];
      stamp();
      stamp();  // (back to back: the cost of a stamp itself)
      V2_FENCE();
    }
#pragma unroll
    for (int ot = 0; ot < Cfg::OT3; ++ot) flush(ot, cur);
    __builtin_amdgcn_wave_barrier();  // (this wave's ctr rows are rewritten by the next unit)
  }
}

// ---- host entry points --------------------------------------------------------------------------------------------
#define SA_DISPATCH(CALL)                                                                   \
  if (C == 1 && c1 == 64 && c2 == 64 && c3 == 64) { CALL(1, 64, 64, 64); }                  \
  else if (C == 64 && c1 == 128 && c2 == 128 && c3 == 256) { CALL(64, 128, 128, 256); }     \
  else {                                                                                    \
    mpx_set_error("mpx_sa (bf16x3): unsupported MLP (C=%d, %d, %d, %d)", C, c1, c2, c3);    \
    return 1;                                                                               \
  }

// Which kernel serves mpx_sa_mlp_bf16x3: the weight-resident one for the first module (pack fits LDS) up to 128 slots
// per neighbourhood, the lockstep one otherwise.  ONE predicate for the launcher and for mpx_sa_mlp_bf16x3_wants_order.
static bool bf16_uses_resident(int C, int c1, int c2, int c3, int nsample) {
  return C == 1 && c1 == 64 && c2 == 64 && c3 == 64 && nsample <= 128;
}

template <int CF, int C1, int C2, int C3>
static int launch_sa_bf16(const float *xyz, int stride, const float *new_xyz, int new_stride, const float *feat,
                          int feat_stride, const int32_t *idx, const int32_t *cnt, const int32_t *order, int B, int N,
                          int npoint, int nsample,
                          const void *wpack, float *out, int out_stride, int append_centre, mpx_stream_t stream) {
  const int64_t nq = (int64_t)B * npoint;
  MPX_REQUIRE(nq < ((int64_t)1 << 31), "mpx_sa_mlp_bf16x3: too many query points");
  constexpr int Q = CF == 1 ? 16 : 4;  // queries per wave
  if constexpr (CF == 1) {
    if (bf16_uses_resident(CF, C1, C2, C3, nsample)) {  // (walks the queries in their natural order: `order` is not needed)
      const int64_t per_wg = (int64_t)res::WV * Q;
      const int wpe = (npoint % per_wg == 0 && B % 8 == 0) ? (int)(npoint / per_wg) : 0;
      const int row16 = (feat == xyz + 3 && stride == 4 && feat_stride == 4 && ((uintptr_t)xyz & 15) == 0) ? 1 : 0;
      const dim3 grid((unsigned)((nq + per_wg - 1) / per_wg));
      const unsigned char *wp = static_cast<const unsigned char *>(wpack);
      if (npoint % Q == 0)  // a wave's queries share their environment
        hipLaunchKernelGGL((sa_mlp_bf16_resident_kernel<CF, C1, C2, C3, Q, true>), grid, dim3(64 * res::WV), 0, mpx_s(stream),
                           xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, nq, N, npoint, nsample, wp, out,
                           out_stride, wpe, row16, append_centre);
      else
        hipLaunchKernelGGL((sa_mlp_bf16_resident_kernel<CF, C1, C2, C3, Q, false>), grid, dim3(64 * res::WV), 0, mpx_s(stream),
                           xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, nq, N, npoint, nsample, wp, out,
                           out_stride, wpe, row16, append_centre);
      MPX_LAUNCH_CHECK("mpx_sa_mlp_bf16x3");
    }
  }
  MPX_REQUIRE(!append_centre, "mpx_sa_mlp_bf16x3: append_centre is implemented by the weight-resident kernel only");
  const int64_t per_block = (int64_t)WAVES * Q;
  hipLaunchKernelGGL((sa_mlp_bf16_kernel<CF, C1, C2, C3, Q>), dim3((unsigned)((nq + per_block - 1) / per_block)),
                     dim3(64 * WAVES), 0, mpx_s(stream), xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, order, nq, N, npoint, nsample,
                     static_cast<const unsigned char *>(wpack), out, out_stride);
  MPX_LAUNCH_CHECK("mpx_sa_mlp_bf16x3");
}

// The factored form has ONE kernel (persistent, barrier-free, its own device-side unit queue): `order` is never needed.
// (Kept in the ABI: callers written against version 200 ask before they sort.)
MPX_EXPORT int mpx_sa_mlp_bf16x3_factored_wants_order(void) { return 0; }

static long long *g_sa2_probe = nullptr;
MPX_EXPORT int mpx_sa_mlp_bf16x3_factored(const float *pre, const float *ctr, const int32_t *idx, const int32_t *cnt,
                                          const int32_t *order, int B, int N, int npoint, int nsample, const void *wpack,
                                          int C, int c1, int c2, int c3, float *out, int out_stride, mpx_stream_t stream) {
  MPX_REQUIRE(C == 64 && c1 == 128 && c2 == 128 && c3 == 256,
              "mpx_sa_mlp_bf16x3_factored: built for the (64+3, 128, 128, 256) module (C=%d, %d, %d, %d)", C, c1, c2, c3);
  MPX_REQUIRE(B >= 0 && N >= 1 && npoint >= 0 && nsample > 0, "mpx_sa_mlp_bf16x3_factored: bad size");
  // (the persistent kernel maps a unit's rows to its queries through an LDS byte table of Q * MAX_NSAMPLE rows per wave)
  static_assert(v2::ROWQ * 4 >= v2::Q * v2::MAX_NSAMPLE + 64, "the row map must hold a unit of Q full neighbourhoods");
  MPX_REQUIRE(nsample <= v2::MAX_NSAMPLE, "mpx_sa_mlp_bf16x3_factored: nsample %d exceeds the %d slots per neighbourhood this kernel is built for",
              nsample, v2::MAX_NSAMPLE);
  MPX_REQUIRE(pre && ctr && idx && cnt, "mpx_sa_mlp_bf16x3_factored: NULL operand (hit counts are required)");
  MPX_REQUIRE(out_stride >= c3, "mpx_sa_mlp_bf16x3_factored: bad stride");
  MPX_REQUIRE((((uintptr_t)wpack | (uintptr_t)pre | (uintptr_t)ctr) & 15) == 0,
              "mpx_sa_mlp_bf16x3_factored: operands must be 16-byte aligned");
  if (B == 0 || npoint == 0) return 0;
  const int64_t nq = (int64_t)B * npoint;
  MPX_REQUIRE(nq < ((int64_t)1 << 31), "mpx_sa_mlp_bf16x3_factored: too many query points");
  (void)order;  // the persistent kernel takes its units from a device-side queue: no sorting pass
  static int cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!cus[dev & 63]) {
    int n = 0;
    MPX_REQUIRE(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0,
                "mpx_sa_mlp_bf16x3_factored: cannot query the CU count");
    cus[dev & 63] = n;
  }
  const int grid = cus[dev & 63];
  const int xcd_aware = (B % 8 == 0 && grid % 8 == 0 && npoint % v2::Q == 0) ? 1 : 0;
  MPX_LDS_LIMIT_ONCE(sa2_bf16x3_persistent_kernel<false>, v2::LDS_BYTES, "mpx_sa_mlp_bf16x3_factored");
  int exhausted = 0;
  unsigned int *queue = mpx_unit_queue_for(mpx_s(stream), &exhausted);
  MPX_REQUIRE(queue != nullptr, exhausted ? "mpx_sa_mlp_bf16x3_factored: no unit-queue slot left for this stream (256 "
                                            "distinct streams per process; reuse streams)"
                                          : "mpx_sa_mlp_bf16x3_factored: cannot reset the unit queue");
  if (g_sa2_probe) {  // (measurement only: mpx_sa2_bf16x3_set_probe)
    MPX_LDS_LIMIT_ONCE(sa2_bf16x3_persistent_kernel<true>, v2::LDS_BYTES, "mpx_sa_mlp_bf16x3_factored");
    hipLaunchKernelGGL(sa2_bf16x3_persistent_kernel<true>, dim3(grid), dim3(64 * v2::WV), v2::LDS_BYTES, mpx_s(stream), idx, cnt,
                       nq, N, npoint, nsample, static_cast<const unsigned char *>(wpack), out, out_stride, pre, ctr, xcd_aware,
                       queue, g_sa2_probe);
    MPX_LAUNCH_CHECK("mpx_sa_mlp_bf16x3_factored");
  }
  hipLaunchKernelGGL(sa2_bf16x3_persistent_kernel<false>, dim3(grid), dim3(64 * v2::WV), v2::LDS_BYTES, mpx_s(stream), idx, cnt,
                     nq, N, npoint, nsample, static_cast<const unsigned char *>(wpack), out, out_stride, pre, ctr, xcd_aware,
                     queue, (long long *)nullptr);
  MPX_LAUNCH_CHECK("mpx_sa_mlp_bf16x3_factored");
}

// measurement only: route the next launches of the persistent kernel through its PROBE instantiation,
// which writes s_memtime stamps of one wave into `probe` (>= 128 int64; nullptr = off)
MPX_EXPORT int mpx_sa2_bf16x3_set_probe(int64_t *probe) {
  g_sa2_probe = reinterpret_cast<long long *>(probe);
  return 0;
}

MPX_EXPORT int mpx_sa_mlp_bf16x3(const float *xyz, int stride, const float *new_xyz, int new_stride,
                                 const float *feat, int feat_stride, int C, const int32_t *idx, const int32_t *cnt,
                                 const int32_t *order, int B, int N, int npoint, int nsample, const void *wpack, int c1,
                                 int c2, int c3, float *out, int out_stride, int append_centre, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N >= 1 && npoint >= 0, "mpx_sa_mlp_bf16x3: bad size");
  MPX_REQUIRE(!append_centre || out_stride >= c3 + 4, "mpx_sa_mlp_bf16x3: append_centre needs out_stride >= c3 + 4");
  MPX_REQUIRE(nsample > 0 && nsample % 32 == 0, "mpx_sa_mlp_bf16x3: nsample must be a positive multiple of 32");
  MPX_REQUIRE(stride >= 3 && new_stride >= 3 && out_stride >= c3, "mpx_sa_mlp_bf16x3: bad stride");
  MPX_REQUIRE(C == 1 || (feat_stride % 4 == 0 && ((uintptr_t)feat & 15) == 0),
              "mpx_sa_mlp_bf16x3: feature rows must be 16-byte aligned");
  MPX_REQUIRE(((uintptr_t)wpack & 15) == 0, "mpx_sa_mlp_bf16x3: wpack must be 16-byte aligned");
  if (B == 0 || npoint == 0) return 0;
#define CALL(a, b, c, d) \
  return launch_sa_bf16<a, b, c, d>(xyz, stride, new_xyz, new_stride, feat, feat_stride, idx, cnt, order, B, N, npoint, nsample, wpack, out, out_stride, append_centre, stream)
  SA_DISPATCH(CALL)
#undef CALL
}

// 1: mpx_sa_mlp_bf16x3 walks the queries in the caller's `order` for this module (the lockstep kernel: balanced
// workgroups); 0: it ignores `order` (the weight-resident kernel) -- and only then supports append_centre
MPX_EXPORT int mpx_sa_mlp_bf16x3_wants_order(int C, int c1, int c2, int c3, int nsample) {
  return bf16_uses_resident(C, c1, c2, c3, nsample) ? 0 : 1;
}

MPX_EXPORT int64_t mpx_sa_pack_bf16x3_size(int C, int c1, int c2, int c3) {
  if (C == 1 && c1 == 64 && c2 == 64 && c3 == 64) return BCfg<1, 64, 64, 64>::TOTAL_BYTES;
  if (C == 64 && c1 == 128 && c2 == 128 && c3 == 256) return BCfg<64, 128, 128, 256>::TOTAL_BYTES;
  return -1;
}

template <int CF, int C1, int C2, int C3>
static int launch_pack_bf16(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                            const float *b3, void *wpack, mpx_stream_t stream) {
  using Cfg = BCfg<CF, C1, C2, C3>;
  const int64_t n = (int64_t)Cfg::STP * 512 + C1 + C2 + C3;
  hipLaunchKernelGGL(sa_pack_bf16_kernel<Cfg>, dim3(cdiv(n, 256)), dim3(256), 0, mpx_s(stream), w1, b1, w2, b2, w3, b3,
                     C1, C2, C3, static_cast<unsigned char *>(wpack));
  MPX_LAUNCH_CHECK("mpx_sa_pack_bf16x3");
}

MPX_EXPORT int mpx_sa_pack_bf16x3(const float *w1, const float *b1, const float *w2, const float *b2,
                                  const float *w3, const float *b3, int C, int c1, int c2, int c3, void *wpack,
                                  mpx_stream_t stream) {
#define CALL(a, b, c, d) return launch_pack_bf16<a, b, c, d>(w1, b1, w2, b2, w3, b3, wpack, stream)
  SA_DISPATCH(CALL)
#undef CALL
}
