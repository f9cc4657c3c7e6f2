// sa3_front_bf16.hip -- the group-all module's first two layers in the split-bf16 ("bf16x3") arithmetic as ONE kernel:
// 128 rows [xyz2 | f2 | 0] (fp32) of an environment -> Linear + ReLU (272 -> 512) -> Linear + ReLU (512 -> 512) -> rows in the
// PAIRS form (hi / lo bf16) that the last layer's kernel (linear_bf16x3_pairs_kernel<3>: 512 -> 1024 + max over the rows)
// stages by DMA.  Reference: PointnetSAModule(mlp=[256(+3), 512, 512, 1024]), npoint = None
// (/root/reference/mpinets/model.py:377-383).  The layer-by-layer form (linear_bf16x3_kernel + linear_bf16x3_pairs_kernel<1>)
// wrote the [B*128, 512] intermediate as 2.1 GB of pairs and read it back (4.8 GB of traffic for 1.6 ms of a kernel whose
// matrix work is 0.9 ms), and its first layer split fp32 rows once per 128-column tile.
//
// Why this is not a twin of sa3_chain_kernel (activations of 64 rows in LDS, every wave streaming the weights of ITS
// channels from L2): a v_mfma_f32_32x32x16_bf16 product costs 3 x 32 cycles per 16 k-values where the fp32 chain pays 8 x 64,
// so the same 3.7 MB weight stream per 64-row pass would have to arrive 5.3x faster -- 43 B / clk / CU from L2 against the
// 32 B / clk / CU at which the tiled pairs kernel already sits at 0.70 of its pipe.  The weights must be shared by >= 128
// rows, and 128 rows x 512 channels of hi / lo pairs are 256 KB: more than the LDS holds.  So here the ROWS are divided
// among the waves and the activations never leave the registers:
//   * one workgroup = one environment, four waves (one per SIMD, 512 registers), wave w owns rows [32 w, 32 w + 32);
//   * every layer computes H^T = W . X^T (A operand = weights, B operand = activations): the 32 x 32 result tile has the
//     ROW on the lane axis and 16 channels on the register axis -- registers 8 u .. 8 u + 7 of tile T are, after ReLU and the
//     hi / lo split, exactly the B operand of k-step 2 T + u of the next layer (its weights are packed in that channel
//     order), so a layer's output becomes the next layer's input without crossing a lane;
//   * the weights are packed on the device (mpx_sa3_front_bf16x3_pack) as 2 KB units (one 32-channel tile x one 16-k step:
//     hi fragment, lo fragment, 16 bytes per lane) in the order every wave consumes them, and travel global -> LDS by DMA
//     in 16 KB chunks through a four-stage ring: each wave brings a quarter of a chunk, ONE barrier per chunk
//     (24 MFMAs of layer 2 per wave), every weight byte is fetched once per environment (21 B / clk / CU) and read from
//     LDS by the four waves (85 B / clk / CU of lane-linear 16-byte reads);
//   * every product is hi*hi + hi*lo + lo*hi with fp32 accumulation, the accumulators start at the bias.
// Summation order: a 16-k step holds the same 16 channels as the pairs kernels' step but in another position order inside
// the MFMA, and the bias is added first instead of last -- equal to the layer-by-layer form to rounding (1e-7 relative),
// not bit for bit.
#include "common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace sa3f {
constexpr int ROWS = 128, WV = 4;
constexpr int K1 = 272, C1 = 512, C2 = 512;
constexpr int S1 = K1 / 16, S2 = C1 / 16;  // k-steps of layers 1, 2
constexpr int T1 = C1 / 32, T2 = C2 / 32;  // 32-channel output tiles
constexpr int G1 = 2, G2 = 2;              // output tiles that share a sweep over k (two accumulator sets of G tiles: the live registers of layer 2 are 256 of operand + 64 + 32 of fragments)
constexpr int UNIT = 2048, CHUNK_UNITS = 8, CHUNK = UNIT * CHUNK_UNITS, NST = 4;
constexpr int U1 = T1 * S1, U2 = T2 * S2, UNITS = U1 + U2;
static_assert(U1 % CHUNK_UNITS == 0 && UNITS % CHUNK_UNITS == 0 && CHUNK_UNITS % G1 == 0 && CHUNK_UNITS % G2 == 0, "steps never straddle chunks");
constexpr int NCHUNK = UNITS / CHUNK_UNITS;
constexpr int64_t BIAS_OFF = (int64_t)UNITS * UNIT;       // b1 [C1] | b2 [C2] as fp32 behind the units
constexpr int64_t PACK_BYTES = BIAS_OFF + 4 * (C1 + C2);
constexpr int LDS_BYTES = NST * CHUNK + 4 * (C1 + C2);
// output stores step j of layer 2 issues (all of them behind the step's barrier): two per half-tile in steps 0-7 of every group ...
constexpr int l2_stores_of(int j) { return ((j + 2 * S2) % S2 < 2 * G2) ? 2 : 0; }
// ... summed over the CHUNK_UNITS / G2 steps since the chunk before was opened (the opening step's own stores included)
constexpr int l2_stores_since_open(int j) {
  int n = 0;
  for (int k = 1; k <= CHUNK_UNITS / G2; ++k) n += l2_stores_of(j - k);
  return n;
}
// channel of the layer-1 output that position p (= 8 half + e) of k-step s of layer 2 holds
__host__ __device__ constexpr int kperm(int s, int p) { return 32 * (s >> 1) + 16 * (s & 1) + (p & 3) + 8 * ((p >> 2) & 1) + 4 * (p >> 3); }
}  // namespace sa3f

// ---- weight packing: unit u = [hi fragment 1 KB | lo fragment 1 KB], lane l of a fragment = 8 bf16 of output channel
// 32 T + (l & 31), k positions 8 (l >> 5) .. + 7 of step s.  Layer 1: units (g, s, t) -> tile G1 g + t, natural k order
// (zero past the real input width); layer 2: units (g, s, t) -> tile G2 g + t, k order kperm().
__global__ void __launch_bounds__(256)
    sa3f_pack_kernel(const float *__restrict__ w1, int k1_real, const float *__restrict__ b1, const float *__restrict__ w2,
                     const float *__restrict__ b2, unsigned char *__restrict__ pack) {
  using namespace sa3f;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)UNITS * 128) {
    const int64_t i = e - (int64_t)UNITS * 128;
    if (i < C1 + C2) reinterpret_cast<float *>(pack + BIAS_OFF)[i] = i < C1 ? b1[i] : b2[i - C1];
    return;
  }
  const int lane = (int)(e & 63), plane = (int)((e >> 6) & 1);
  int u = (int)(e >> 7);
  const bool l2 = u >= U1;
  if (l2) u -= U1;
  const int G = l2 ? G2 : G1, S = l2 ? S2 : S1;
  const int t = u % G, s = (u / G) % S, g = u / (G * S);
  const int ch = 32 * (G * g + t) + (lane & 31), h = lane >> 5;
  bf16x8 o;
#pragma unroll
  for (int el = 0; el < 8; ++el) {
    float v;
    if (l2) {
      v = w2[(size_t)ch * C1 + kperm(s, 8 * h + el)];
    } else {
      const int k = 16 * s + 8 * h + el;
      v = k < k1_real ? w1[(size_t)ch * k1_real + k] : 0.0f;
    }
    const __bf16 hi = (__bf16)v;
    o[el] = plane ? (__bf16)(v - (float)hi) : hi;
  }
  *reinterpret_cast<bf16x8 *>(pack + (size_t)(e >> 7) * UNIT + plane * 1024 + lane * 16) = o;
}

// the last layer's weights with their columns in the order the kernel below writes its output rows (k-step s, position p
// <- channel kperm(s, p)), as the pairs the pairs kernels take
__global__ void __launch_bounds__(256)
    sa3f_w3_pairs_kernel(const float *__restrict__ w3, int c3, __bf16 *__restrict__ out) {
  using namespace sa3f;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)c3 * C2) return;
  const int n = (int)(e / C2), kk = (int)(e % C2);  // position kk of the permuted row
  const float v = w3[(size_t)n * C2 + kperm(kk >> 4, kk & 15)];
  const __bf16 hi = (__bf16)v;
  const size_t o = (size_t)n * 2 * C2 + (size_t)(kk >> 4) * 32 + (kk & 15);
  out[o] = hi;
  out[o + 16] = (__bf16)(v - (float)hi);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
namespace sa3f {
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
}  // namespace sa3f

template <bool PROBE>
__global__ void __launch_bounds__(64 * sa3f::WV) __attribute__((amdgpu_waves_per_eu(1, 1)))
    sa3_front_bf16x3_kernel(const float *__restrict__ x, int ldx, const unsigned char *__restrict__ pack,
                            __bf16 *__restrict__ yp, int ldp, long long *__restrict__ probe) {
  using namespace sa3f;
  // (PROBE: s_memtime at the phase boundaries, kept in scalar registers and written out by one thread at the very end)
  long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int pi = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (PROBE) ts[pi++] = (long long)__builtin_amdgcn_s_memtime();
  };
  stamp();
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];  // [NST][CHUNK] | biases b1 | b2 (fp32)
  float *bias_s = reinterpret_cast<float *>(ring + NST * CHUNK);
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int64_t row = (int64_t)blockIdx.x * ROWS + wave * 32 + col;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(pack), 0, (int)PACK_BYTES, 0x00020000);
  const int voff = lane * 16;
  // a wave brings its quarter of chunk c: four 1 KB pieces, lane-linear in memory and in LDS
  auto dma = [&](int c, int dyn_bytes) __attribute__((always_inline)) {  // c: compile-time part of the chunk index (stage, base offset)
    unsigned char *dst = ring + (c % NST) * CHUNK + wave * (CHUNK / WV);
    const int src = dyn_bytes + c * CHUNK + wave * (CHUNK / WV);
#pragma unroll
    for (int i = 0; i < CHUNK / WV / 1024; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(dst + i * 1024), 16, voff, src + i * 1024, 0, 0);
  };
  constexpr int LPC = CHUNK / WV / 1024;  // DMA loads per wave and chunk


  // ---- biases -> LDS; this lane's half of its input row: k-step s takes floats 16 s + 8 half .. + 7 (requested before the
  // first weights: the counted wait in front of the first barrier then covers them)
  bf16x8 xh[S1], xl[S1];
  {
    const float4 bq = reinterpret_cast<const float4 *>(pack + BIAS_OFF)[tid];  // (C1 + C2) / 4 = 256 = one per thread
    static_assert((C1 + C2) / 4 == 64 * WV, "one 16-byte piece of the biases per thread");
    const float *xr = x + row * ldx + 8 * half;
    float4 raw[S1][2];
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      raw[s][0] = *reinterpret_cast<const float4 *>(xr + 16 * s);
      raw[s][1] = *reinterpret_cast<const float4 *>(xr + 16 * s + 4);
    }
    dma(0, 0);
    dma(1, 0);
    reinterpret_cast<float4 *>(bias_s)[tid] = bq;
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      const float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
      split8(v, xh[s], xl[s]);
    }
  }
  // Chunk c becomes readable behind barrier B(c), which every wave passes with its own quarter landed; B(c) stands in front
  // of the first fragment read of the chunk, i.e. one step before its first MFMA.  Passing B(c) also says that every wave is
  // done with chunk c - 2, so chunk c + 2 is requested into that stage right behind the barrier: two chunks (1.5 k matrix
  // cycles) of latency cover.  Counted wait: vector-memory operations complete in issue order, and in front of B(c) the
  // younger ones are the LPC loads of chunk c + 1 plus the `younger_stores` output stores issued since (every store sits at
  // a fixed place of this fully unrolled stream, between two side-effecting builtins it cannot be moved across) -- waiting
  // for the stores too (as a plain vmcnt(LPC) does) held every wave for ~10 k cycles behind each group's 64 scattered stores.
  // Chunks past the end of the pack are requested like the others (the descriptor's range check answers zeros into a stage
  // nobody reads any more): every barrier then sees the same number of younger loads, also inside the rolled loop of layer 2.
  auto open_chunk = [&](auto C, auto YS, int dyn_bytes) __attribute__((always_inline)) {
    constexpr int c = decltype(C)::value, younger_stores = decltype(YS)::value;
    constexpr int n = LPC + younger_stores;
    static_assert(n <= 63, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));  // vmcnt(n); lgkmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    dma(c + 2, dyn_bytes);
  };
  // fragments of G consecutive units starting at unit u0 (compile-time: stage and offsets are immediates of the reads)
  auto read_frags = [&](auto U0, auto Gt, bf16x8 (&fh)[G2], bf16x8 (&fl)[G2]) __attribute__((always_inline)) {
    constexpr int u0 = decltype(U0)::value, G = decltype(Gt)::value;
#pragma unroll
    for (int t = 0; t < G; ++t) {
      const int u = u0 + t;
      const unsigned char *p = ring + ((u / CHUNK_UNITS) % NST) * CHUNK + (u % CHUNK_UNITS) * UNIT + voff;
      fh[t] = *reinterpret_cast<const bf16x8 *>(p);
      fl[t] = *reinterpret_cast<const bf16x8 *>(p + 1024);
    }
  };
  // half u of accumulator tile T (registers 8 u .. 8 u + 7 = channels 32 T + 16 u + (e & 3) + 8 (e >> 2) + 4 half) + bias,
  // ReLU, split: the B operand of k-step 2 T + u of the next layer
  auto finish_half = [&](const f32x16 &acc, const float *b, int T, int u, bf16x8 &hi, bf16x8 &lo) __attribute__((always_inline)) {
    const float4 b0 = *reinterpret_cast<const float4 *>(b + 32 * T + 16 * u + 4 * half);
    const float4 b1 = *reinterpret_cast<const float4 *>(b + 32 * T + 16 * u + 8 + 4 * half);
    const float v[8] = {fmaxf(acc[8 * u + 0] + b0.x, 0.0f), fmaxf(acc[8 * u + 1] + b0.y, 0.0f), fmaxf(acc[8 * u + 2] + b0.z, 0.0f),
                        fmaxf(acc[8 * u + 3] + b0.w, 0.0f), fmaxf(acc[8 * u + 4] + b1.x, 0.0f), fmaxf(acc[8 * u + 5] + b1.y, 0.0f),
                        fmaxf(acc[8 * u + 6] + b1.z, 0.0f), fmaxf(acc[8 * u + 7] + b1.w, 0.0f)};
    split8(v, hi, lo);
  };
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  bf16x8 fh[2][G2], fl[2][G2];  // weight fragments of the current and the next step
  stamp();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this thread's piece of the biases is in LDS before the first barrier
  open_chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
  stamp();
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, G1>{}, fh[0], fl[0]);

  // (one scheduling region per step: the compiler interleaves the step's reads, MFMAs and deferred output work itself;
  // sched_group_barrier patterns over this stream did not compile in finite time)
  auto step_order = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  // ---- layer 1: 8 groups of 2 tiles x 17 k-steps; steps j = g S1 + s, units 2 j, 2 j + 1.  The output of group g (bias,
  // ReLU, split: 4 half-tiles) is finished during steps 0-3 of group g + 1 (of layer 2's first group for the last one) from
  // the other accumulator set, so the matrix pipe does not wait for it.
  bf16x8 h1h[S2], h1l[S2];  // layer-1 output = layer-2 operand, k-step 2 T + u <- registers 8 u .. 8 u + 7 of tile T
  f32x16 acc1[2][G1];
  {
    static_for<0, T1 / G1 * S1>([&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value, g = j / S1, s = j % S1, buf = j & 1, cur = g & 1;
      constexpr int u_next = (j + 1) * G1;  // first unit of the next step (the first step of layer 2 behind the last one)
      if constexpr (u_next % CHUNK_UNITS == 0 && u_next < UNITS)
        open_chunk(std::integral_constant<int, u_next / CHUNK_UNITS>{}, std::integral_constant<int, 0>{}, 0);
      if constexpr (u_next < U1) read_frags(std::integral_constant<int, u_next>{}, std::integral_constant<int, G1>{}, fh[buf ^ 1], fl[buf ^ 1]);
      else read_frags(std::integral_constant<int, u_next>{}, std::integral_constant<int, G2>{}, fh[buf ^ 1], fl[buf ^ 1]);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < G1; ++t)
          acc1[cur][t] = mfma(p == 1 ? fl[buf][t] : fh[buf][t], p == 2 ? xl[s] : xh[s], (s == 0 && p == 0) ? zero : acc1[cur][t]);
      if constexpr (g >= 1 && s < 2 * G1) {
        constexpr int T = G1 * (g - 1) + (s >> 1), u = s & 1;
        finish_half(acc1[cur ^ 1][s >> 1], bias_s, T, u, h1h[2 * T + u], h1l[2 * T + u]);
      }
      step_order();
    });
  }
  stamp();
  // ---- layer 1's last group: finished here (four half-tiles; their k-steps 28-31 are the last ones layer 2 needs) --------
  {
    constexpr int GL1 = T1 / G1 - 1;
#pragma unroll
    for (int e = 0; e < 2 * G1; ++e) {
      const int T = G1 * GL1 + (e >> 1), u = e & 1;
      finish_half(acc1[GL1 & 1][e >> 1], bias_s, T, u, h1h[2 * T + u], h1l[2 * T + u]);
    }
  }
  // ---- layer 2: 4 groups of 4 tiles x 32 k-steps; the output rows leave in the pairs form, k-step order = this kernel's
  // register order (the last layer's weight columns are permuted to match: sa3f_w3_pairs_kernel); a group's 8 half-tiles are
  // finished and stored (two 16-byte stores each) during steps 0-7 of the next group.  The groups run as a ROLLED loop of two
  // groups per iteration (one per accumulator set): fully unrolled the kernel was 85 KB of code executed once per workgroup
  // -- larger than the instruction cache, and every variant of it (no weight traffic, no LDS reads, no stores, no barriers)
  // took the same 64 cycles per MFMA; layer 1 must stay unrolled (its outputs are registers indexed by the group).
  {
    constexpr int J0 = T1 / G1 * S1;      // (buffer parity continues from layer 1)
    constexpr int NG2 = T2 / G2;          // groups
    static_assert(NG2 % 2 == 0 && (2 * S2 * G2) % (CHUNK_UNITS * NST) == 0, "an iteration = two groups = a whole number of ring turns");
    f32x16 acc2[2][G2];
#pragma unroll
    for (int t = 0; t < G2; ++t) acc2[1][t] = zero;  // (read by the first group's placeholder stores, see below)
    __bf16 *yrow = yp + row * (int64_t)ldp + 8 * half;
    auto store_half = [&](int T, int u, const bf16x8 &hi, const bf16x8 &lo) __attribute__((always_inline)) {
      __bf16 *dst = yrow + (2 * T + u) * 32;  // k-step 2 T + u of the output row: [hi x 16 | lo x 16], this lane-half's 8 of each
      *reinterpret_cast<bf16x8 *>(dst) = hi;
      *reinterpret_cast<bf16x8 *>(dst + 16) = lo;
    };
#pragma nounroll
    for (int it = 0; it < NG2 / 2; ++it) {
      const int dyn = it * (2 * S2 * G2) * UNIT;  // bytes of weights behind the iterations before this one
      static_for<0, 2 * S2>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, gg = j / S2, s = j % S2, buf = (J0 + j) & 1;
        constexpr int u_next = U1 + (j + 1) * G2;  // (of iteration 0; the chunk it starts is chunk u_next / CHUNK_UNITS + 32 it)
        // every group stores its predecessor's 8 half-tiles in its steps 0-7: two stores per step, behind the step's barrier
        if constexpr (u_next % CHUNK_UNITS == 0)
          open_chunk(std::integral_constant<int, u_next / CHUNK_UNITS>{}, std::integral_constant<int, l2_stores_since_open(j)>{}, dyn);
        if constexpr (j + 1 < 2 * S2) {
          read_frags(std::integral_constant<int, u_next>{}, std::integral_constant<int, G2>{}, fh[buf ^ 1], fl[buf ^ 1]);
        } else {  // the first step of the next iteration: the same stages one ring turn later (past the end: never used)
          read_frags(std::integral_constant<int, U1>{}, std::integral_constant<int, G2>{}, fh[buf ^ 1], fl[buf ^ 1]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int t = 0; t < G2; ++t)
            acc2[gg][t] = mfma(p == 1 ? fl[buf][t] : fh[buf][t], p == 2 ? h1l[s] : h1h[s], (s == 0 && p == 0) ? zero : acc2[gg][t]);
        if constexpr (s < 2 * G2) {
          // the group before (2 it + gg - 1; for the very first group the LAST one's place: placeholder rows that the real
          // results overwrite behind the loop -- same lane, same addresses, later in program order)
          const int gp = (2 * it + gg + NG2 - 1) % NG2, T = G2 * gp + (s >> 1);
          bf16x8 hi, lo;
          finish_half(acc2[gg ^ 1][s >> 1], bias_s + C1, T, s & 1, hi, lo);
          store_half(T, s & 1, hi, lo);
        }
        step_order();
      });
    }
    stamp();
    constexpr int GL = NG2 - 1;
#pragma unroll
    for (int e = 0; e < 2 * G2; ++e) {
      bf16x8 hi, lo;
      finish_half(acc2[GL & 1][e >> 1], bias_s + C1, G2 * GL + (e >> 1), e & 1, hi, lo);
      store_half(G2 * GL + (e >> 1), e & 1, hi, lo);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (the requests past the end of the pack have landed before the wave leaves)
    stamp();
  }
  if constexpr (PROBE) {
    if (blockIdx.x == 300 && threadIdx.x == 0)
      for (int i = 0; i < 8; ++i) probe[i] = ts[i];
    if (threadIdx.x == 0) {  // every workgroup: first stamp, last stamp, where it ran (probe[64 + 4 b ...])
      probe[64 + 4 * blockIdx.x + 0] = ts[0];
      probe[64 + 4 * blockIdx.x + 1] = ts[pi - 1];
      probe[64 + 4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
      probe[64 + 4 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
  }
}

// ---- host entry points ----------------------------------------------------------------------------------------------
MPX_EXPORT int64_t mpx_sa3_front_bf16x3_pack_size(int K1, int c1, int c2) {
  return (K1 == sa3f::K1 && c1 == sa3f::C1 && c2 == sa3f::C2) ? sa3f::PACK_BYTES : -1;
}

MPX_EXPORT int mpx_sa3_front_bf16x3_pack(const float *w1, int k1_real, const float *b1, const float *w2, const float *b2, int K1,
                                         int c1, int c2, void *pack, mpx_stream_t stream) {
  MPX_REQUIRE(K1 == sa3f::K1 && c1 == sa3f::C1 && c2 == sa3f::C2, "mpx_sa3_front_bf16x3_pack: built for (272 -> 512 -> 512), got (%d, %d, %d)",
              K1, c1, c2);
  MPX_REQUIRE(w1 && b1 && w2 && b2 && pack && k1_real >= 1 && k1_real <= K1, "mpx_sa3_front_bf16x3_pack: bad operand");
  MPX_REQUIRE(((uintptr_t)pack & 15) == 0, "mpx_sa3_front_bf16x3_pack: pack must be 16-byte aligned");
  const int64_t n = (int64_t)sa3f::UNITS * 128 + sa3f::C1 + sa3f::C2;
  hipLaunchKernelGGL(sa3f_pack_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, mpx_s(stream), w1, k1_real, b1, w2, b2,
                     static_cast<unsigned char *>(pack));
  MPX_LAUNCH_CHECK("mpx_sa3_front_bf16x3_pack");
}

MPX_EXPORT int mpx_sa3_front_bf16x3_w3_pairs(const float *w3, int c3, int c2, void *w3_pairs, mpx_stream_t stream) {
  MPX_REQUIRE(c2 == sa3f::C2 && c3 >= 1 && w3 && w3_pairs, "mpx_sa3_front_bf16x3_w3_pairs: bad operand (c2 must be %d)", sa3f::C2);
  hipLaunchKernelGGL(sa3f_w3_pairs_kernel, dim3((unsigned)cdiv((int64_t)c3 * c2, 256)), dim3(256), 0, mpx_s(stream), w3, c3,
                     static_cast<__bf16 *>(w3_pairs));
  MPX_LAUNCH_CHECK("mpx_sa3_front_bf16x3_w3_pairs");
}

MPX_EXPORT int mpx_sa3_front_bf16x3(const float *x, int ldx, int B, int rows, const void *pack, void *y_pairs, int ldp,
                                    mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && x && pack && y_pairs, "mpx_sa3_front_bf16x3: bad operand");
  MPX_REQUIRE(rows == sa3f::ROWS, "mpx_sa3_front_bf16x3: an environment has exactly %d rows (got %d)", sa3f::ROWS, rows);
  MPX_REQUIRE(ldx >= sa3f::K1 && ldx % 4 == 0 && ldp >= 2 * sa3f::C2 && ldp % 8 == 0, "mpx_sa3_front_bf16x3: bad leading dimension");
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)pack | (uintptr_t)y_pairs) & 15) == 0, "mpx_sa3_front_bf16x3: operands must be 16-byte aligned");
  if (B == 0) return 0;
  MPX_LDS_LIMIT_ONCE(sa3_front_bf16x3_kernel<false>, sa3f::LDS_BYTES, "mpx_sa3_front_bf16x3");
  hipLaunchKernelGGL(sa3_front_bf16x3_kernel<false>, dim3((unsigned)B), dim3(64 * sa3f::WV), sa3f::LDS_BYTES, mpx_s(stream), x, ldx,
                     static_cast<const unsigned char *>(pack), static_cast<__bf16 *>(y_pairs), ldp, (long long *)nullptr);
  MPX_LAUNCH_CHECK("mpx_sa3_front_bf16x3");
}

// measurement only (tools/probes/sa3_front_probe.py): the same launch with s_memtime stamps of workgroup 300, wave 0 at
// the phase boundaries -> probe[0 .. 32)
MPX_EXPORT int mpx_sa3_front_bf16x3_probe(const float *x, int ldx, int B, const void *pack, void *y_pairs, int ldp, int64_t *probe,
                                          mpx_stream_t stream) {
  MPX_REQUIRE(B > 300 && x && pack && y_pairs && probe, "mpx_sa3_front_bf16x3_probe: bad operand (B > 300)");
  MPX_LDS_LIMIT_ONCE(sa3_front_bf16x3_kernel<true>, sa3f::LDS_BYTES, "mpx_sa3_front_bf16x3_probe");
  hipLaunchKernelGGL(sa3_front_bf16x3_kernel<true>, dim3((unsigned)B), dim3(64 * sa3f::WV), sa3f::LDS_BYTES, mpx_s(stream), x, ldx,
                     static_cast<const unsigned char *>(pack), static_cast<__bf16 *>(y_pairs), ldp,
                     reinterpret_cast<long long *>(probe));
  MPX_LAUNCH_CHECK("mpx_sa3_front_bf16x3_probe");
}
