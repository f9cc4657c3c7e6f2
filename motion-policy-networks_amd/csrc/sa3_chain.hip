// sa3_chain.hip -- the group-all set-abstraction module as ONE kernel: 128 rows [xyz2 | f2 | 0] of an environment ->
// Linear + ReLU (K1 -> C1) -> Linear + ReLU (C1 -> C2) -> Linear + ReLU (C2 -> C3) -> max over the 128 rows.
//
// Reference: PointnetSAModule(mlp=[256(+3), 512, 512, 1024]) with npoint = None (/root/reference/mpinets/model.py:377-383):
// GroupAll + three 1x1 convolutions + max over the points.  The engine used to run it as three GEMMs over B*128 rows
// (mpx_linear, mpx_linear, mpx_linear_rowmax): two [B*128, 512] fp32 intermediates were written and read back (4.3 GB
// each way per step at 8192 environments).  Here nothing but the input rows and the pooled [1024] row touches HBM.
//
// CDNA4 mapping (exact-fp32 matrix cores, v_mfma_f32_32x32x2_f32; 157 TFLOP/s peak):
//   * one workgroup (4 waves, one per SIMD) = one environment, walked as two passes of 64 rows;
//   * the activations of the pass live in LDS as H[64 rows][516] (row stride padded by 4 floats: the 16-byte operand reads
//     of 8 consecutive rows fall into 8 different 16-byte bank groups);
//   * wave w owns output channels [128 w, 128 w + 128) of layers 1-2 (256 of layer 3, in two halves): eight independent
//     32x32 accumulators (4 channel tiles x 2 row tiles), so consecutive MFMAs never wait for each other;
//   * layers 1-2 compute H^T = W . X^T (A operand = weights, B operand = activations): the result tile has the ROW on the
//     lane axis and four CONSECUTIVE channels in registers 4j .. 4j+3 -- it goes back to LDS as 16-byte stores and is
//     exactly what the next layer reads as its B operand (k-step s of a 32-channel tile takes channel
//     (s & 3) + 8 (s >> 2) + 4 half: a lane's four k-steps are one float4);
//   * the last layer flips roles (A = activations, B = weights): rows land on the register axis, the max over the rows is
//     an in-lane max over 16 registers x 2 row tiles + one cross-half exchange; bias + ReLU after the max (they commute);
//   * weights are packed on the device (mpx_sa3_pack_weights) in the order the waves consume them: 16 bytes per lane per 4
//     k-steps, each wave a contiguous stream, requested three groups ahead through a four-stage register ring; the whole pack
//     (3.6 MB) is L2-resident and every workgroup streams it once per 64-row pass (7.8 B / clk / CU).
// Per pass a wave issues 1088 + 2048 + 4096 MFMAs (463 k matrix cycles); the non-matrix work between layers (operand
// write-back, staging of the next 64 input rows, 6 barriers) is ~2 % of that.
// Summation order: k walks the channel order above inside each 32-channel tile -- a different fp32 rounding order than
// the plain GEMM kernels (1e-7 relative), the same for every batch size this kernel serves.
#include "common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace sa3 {
constexpr int ROWS = 128, PR = 64, WV = 4;

template <int K1, int C1, int C2, int C3>
struct Cfg {
  static_assert(K1 % 8 == 0 && K1 % 16 == 0, "input rows are whole 16-float slabs");
  static_assert(C1 == 128 * WV && C2 == 128 * WV && C3 == 256 * WV, "a wave owns 128 channels (256 of the last layer)");
  static constexpr int KG1 = K1 / 8, KG2 = C1 / 8, KG3 = C2 / 8;  // groups of 4 k-steps (8 channels: 4 per lane half)
  static constexpr int64_t W1_OFF = 0, W2_OFF = (int64_t)C1 * K1, W3_OFF = W2_OFF + (int64_t)C2 * C1;
  static constexpr int64_t B1_OFF = W3_OFF + (int64_t)C3 * C2, B2_OFF = B1_OFF + C1, B3_OFF = B2_OFF + C2;
  static constexpr int64_t TOTAL = B3_OFF + C3;
  static constexpr int KMAX = K1 > C1 ? (K1 > C2 ? K1 : C2) : (C1 > C2 ? C1 : C2);
  static constexpr int LD = KMAX + 4;
  static constexpr int LDS_BYTES = PR * LD * 4;
};
}  // namespace sa3

// ---- weight packing -------------------------------------------------------------------------------------------------
// stream element (float4) index of layer L: ((unit * KG + g) * 4 + ot) * 64 + lane, unit = wave (layers 1-2) or
// wave * 2 + half-of-its-256-channels (layer 3); the float4 holds W[out = unit * 128 + ot * 32 + (lane & 31)]
// [in = 8 g + 4 (lane >> 5) + 0..3] (zero past the real input width: K1 is padded).
template <class C>
__global__ void __launch_bounds__(256)
    sa3_pack_kernel(const float *__restrict__ w1, int k1_real, const float *__restrict__ b1, const float *__restrict__ w2,
                    const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3, int K1, int C1,
                    int C2, int C3, float *__restrict__ pack) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= C::TOTAL) return;
  float v;
  if (e >= C::B1_OFF) {
    const int64_t i = e - C::B1_OFF;
    v = i < C1 ? b1[i] : (i < C1 + C2 ? b2[i - C1] : b3[i - C1 - C2]);
  } else {
    const float *w;
    int kin, kreal, KG;
    int64_t r = e;
    if (r >= C::W3_OFF) { r -= C::W3_OFF; w = w3; kin = C2; kreal = C2; KG = C::KG3; }
    else if (r >= C::W2_OFF) { r -= C::W2_OFF; w = w2; kin = C1; kreal = C1; KG = C::KG2; }
    else { w = w1; kin = K1; kreal = k1_real; KG = C::KG1; }
    const int i = (int)(r & 3), lane = (int)((r >> 2) & 63), ot = (int)((r >> 8) & 3);
    const int64_t ug = r >> 10;  // unit * KG + g
    const int g = (int)(ug % KG), unit = (int)(ug / KG);
    const int out = unit * 128 + ot * 32 + (lane & 31), in = 8 * g + 4 * (lane >> 5) + i;
    (void)kin;
    v = in < kreal ? w[(size_t)out * kreal + in] : 0.0f;
  }
  pack[e] = v;
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 sa3_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4 sa3_bload16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}
__device__ __forceinline__ float sa3_comp(const float4 &q, int i) { return i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w)); }

template <int K1, int C1, int C2, int C3, bool PROBE = false>
__global__ void __launch_bounds__(64 * sa3::WV) __attribute__((amdgpu_waves_per_eu(1, 1)))
    sa3_chain_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ pack, float *__restrict__ out, int ldo,
                     long long *__restrict__ probe = nullptr) {
  using C = sa3::Cfg<K1, C1, C2, C3>;
  using namespace sa3;
  extern __shared__ __attribute__((aligned(16))) float H[];  // [PR][LD]
  constexpr int LD = C::LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int64_t env = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pack), 0, (int)(C::TOTAL * 4), 0x00020000);
  const int voff = lane * 16;
  const float *hrow0 = H + col * LD + 4 * half;          // this lane's operand row of row tile 0 (+ 32 LD: row tile 1)
  float pool[2][4];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) pool[hf][ot] = -__builtin_inff();

  // one layer's matrix work for this wave: KG groups of 4 k-steps; per group 4 weight float4 (one per channel tile) from
  // the stream at `wbase` and 2 activation float4 (one per row tile) from LDS; 32 MFMAs on 8 accumulators.
  // FLIP: A = weights, B = activations (layers 1-2); else A = activations, B = weights (last layer).
  auto run = [&](auto KGt, auto FLIPt, int wbase, f32x16 (&acc)[4][2]) __attribute__((always_inline)) {
    constexpr int KG = decltype(KGt)::value;
    constexpr bool FLIP = decltype(FLIPt)::value;
    // operand ring: NS stages, a group's operands are requested NS - 1 groups (96 MFMAs, ~6 k matrix cycles) before they
    // are used -- the 3.6 MB pack competes with the activation rows for a 4 MB L2 and part of the stream comes from
    // the Infinity Cache (measured with a two-stage ring: the matrix pipes waited ~18 % of the time on it)
    constexpr int NS = 4;
    float4 wr[NS][4], br[NS][2];
    auto fetch = [&](int st, int g) __attribute__((always_inline)) {
      const int soff = wbase + g * 4096;  // (one scalar add per group: the channel tile rides in the instruction's offset field)
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) wr[st][ot] = sa3_bload16(rsrc, voff + ot * 1024, soff);
      br[st][0] = *reinterpret_cast<const float4 *>(hrow0 + 8 * g);
      br[st][1] = *reinterpret_cast<const float4 *>(hrow0 + 32 * LD + 8 * g);
    };
    auto compute = [&](int st) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const float w = sa3_comp(wr[st][ot], i), a = sa3_comp(br[st][rt], i);
            acc[ot][rt] = FLIP ? sa3_mfma(w, a, acc[ot][rt]) : sa3_mfma(a, w, acc[ot][rt]);
          }
    };
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) fetch(st, st);
    // issue order of a group: ONE memory instruction per gap between MFMAs.  A lone wave on a SIMD hides a few issue
    // slots behind each 64-cycle MFMA but pays for a cluster: with the four weight loads and the two LDS reads of a
    // group issued back to back the layer loops ran at 0.90 of the matrix floor (s_memtime probe).
    auto issue_order = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one weight load
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS operand read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    // full rounds: every group of the round requests the group NS - 1 ahead (no conditions inside the loop: the wait
    // counters stay exact); the last groups are unrolled with compile-time conditions
    constexpr int ROUNDS = (KG - NS + 1) / NS;
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        fetch((st + NS - 1) % NS, r * NS + st + NS - 1);
        compute(st);
        issue_order();
      }
    }
    static_for<ROUNDS * NS, KG>([&](auto G) __attribute__((always_inline)) {
      constexpr int g = decltype(G)::value;
      if constexpr (g + NS - 1 < KG) {
        fetch((g + NS - 1) % NS, g + NS - 1);
        compute(g % NS);
        issue_order();
      } else {
        compute(g % NS);
      }
    });
  };
  // accumulators of a flipped layer start at the bias: register 4 j + i of channel tile ot = channel 32 ot + 8 j + 4 half + i
  auto bias_init = [&](int boff_floats, f32x16 (&acc)[4][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b = sa3_bload16(rsrc, half * 16, (boff_floats + wave * 128 + ot * 32 + 8 * j) * 4);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          acc[ot][rt][4 * j + 0] = b.x, acc[ot][rt][4 * j + 1] = b.y, acc[ot][rt][4 * j + 2] = b.z, acc[ot][rt][4 * j + 3] = b.w;
        }
      }
  };
  // relu(accumulators) -> LDS as the next layer's operand rows: lane (row, half) owns channels 8 j + 4 half .. + 3 of a tile
  auto write_back = [&](const f32x16 (&acc)[4][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v;
          v.x = fmaxf(acc[ot][rt][4 * j + 0], 0.0f), v.y = fmaxf(acc[ot][rt][4 * j + 1], 0.0f);
          v.z = fmaxf(acc[ot][rt][4 * j + 2], 0.0f), v.w = fmaxf(acc[ot][rt][4 * j + 3], 0.0f);
          *reinterpret_cast<float4 *>(H + (rt * 32 + col) * LD + wave * 128 + ot * 32 + 8 * j + 4 * half) = v;
        }
  };

  int pi = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (PROBE) {
      if (blockIdx.x == 300 && tid == 0) probe[pi] = (long long)__builtin_amdgcn_s_memtime();
      ++pi;
    }
  };
  stamp();
  for (int pass = 0; pass < ROWS / PR; ++pass) {
    // ---- the pass's 64 input rows -> LDS (coalesced 16-byte copies; the buffer is free: the previous pass's last layer
    // has been read by every wave -- barrier at the end of the pass)
    // (all of a thread's 17 loads are in flight before the first LDS store: as a rolled load -> wait -> store loop the
    // staging took 20 k cycles per pass, 4 % of the kernel -- s_memtime probe)
    const float *xr = x + (env * ROWS + pass * PR) * (int64_t)ldx;
    constexpr int NX = PR * (K1 / 4) / (64 * WV);
    static_assert(NX * 64 * WV == PR * (K1 / 4), "the input rows split evenly over the threads");
    {
      float4 xb[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int i = tid + j * 64 * WV, r = i / (K1 / 4), c4 = i - r * (K1 / 4);
        xb[j] = *reinterpret_cast<const float4 *>(xr + (int64_t)r * ldx + 4 * c4);
      }
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int i = tid + j * 64 * WV, r = i / (K1 / 4), c4 = i - r * (K1 / 4);
        *reinterpret_cast<float4 *>(H + r * LD + 4 * c4) = xb[j];
      }
    }
    __syncthreads();
    stamp();
    f32x16 acc[4][2];
    // ---- layer 1: K1 -> C1
    bias_init((int)C::B1_OFF, acc);
    run(std::integral_constant<int, C::KG1>{}, std::true_type{}, (int)(C::W1_OFF * 4) + wave * C::KG1 * 4096, acc);
    stamp();
    __syncthreads();  // every wave has read the input rows
    write_back(acc);
    __syncthreads();
    stamp();
    // ---- layer 2: C1 -> C2
    bias_init((int)C::B2_OFF, acc);
    run(std::integral_constant<int, C::KG2>{}, std::true_type{}, (int)(C::W2_OFF * 4) + wave * C::KG2 * 4096, acc);
    stamp();
    __syncthreads();
    write_back(acc);
    __syncthreads();
    stamp();
    // ---- layer 3: C2 -> C3 in two halves of this wave's 256 channels; roles flipped; pooled over the pass's rows
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[ot][rt] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      run(std::integral_constant<int, C::KG3>{}, std::false_type{}, (int)(C::W3_OFF * 4) + (wave * 2 + hf) * C::KG3 * 4096, acc);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        float m = fmaxf(acc[ot][0][0], acc[ot][1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, fmaxf(acc[ot][0][r], acc[ot][1][r]));
        pool[hf][ot] = fmaxf(pool[hf][ot], m);
      }
      stamp();
    }
    __syncthreads();  // the last layer's operand rows are dead: the next pass may overwrite them
    stamp();
  }
  // ---- pooled row: the two lane halves hold disjoint rows; bias + ReLU after the max
  const float *b3 = pack + C::B3_OFF;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
      const float v = mpx_max_across_halves(pool[hf][ot]);
      const int ch = wave * 256 + hf * 128 + ot * 32 + col;
      if (half == 0) out[env * (int64_t)ldo + ch] = fmaxf(v + b3[ch], 0.0f);
    }
}

// ---- host entry points ----------------------------------------------------------------------------------------------
#define SA3_DISPATCH(CALL)                                                                              \
  if (K1 == 272 && c1 == 512 && c2 == 512 && c3 == 1024) { CALL(272, 512, 512, 1024); }                 \
  else {                                                                                                \
    mpx_set_error("mpx_sa3: unsupported group-all MLP (K1=%d, %d, %d, %d)", K1, c1, c2, c3);            \
    return 1;                                                                                           \
  }

MPX_EXPORT int64_t mpx_sa3_pack_size(int K1, int c1, int c2, int c3) {
  if (K1 == 272 && c1 == 512 && c2 == 512 && c3 == 1024) return sa3::Cfg<272, 512, 512, 1024>::TOTAL;
  return -1;
}

MPX_EXPORT int mpx_sa3_pack_weights(const float *w1, int k1_real, const float *b1, const float *w2, const float *b2,
                                    const float *w3, const float *b3, int K1, int c1, int c2, int c3, float *pack,
                                    mpx_stream_t stream) {
  MPX_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && pack, "mpx_sa3_pack_weights: NULL operand");
  MPX_REQUIRE(k1_real >= 1 && k1_real <= K1, "mpx_sa3_pack_weights: the real input width %d must be in [1, K1 = %d]", k1_real, K1);
#define CALL(a, b, c, d)                                                                                             \
  do {                                                                                                               \
    using C = sa3::Cfg<a, b, c, d>;                                                                                  \
    hipLaunchKernelGGL(sa3_pack_kernel<C>, dim3(cdiv(C::TOTAL, 256)), dim3(256), 0, mpx_s(stream), w1, k1_real, b1, w2, \
                       b2, w3, b3, K1, c1, c2, c3, pack);                                                            \
    MPX_LAUNCH_CHECK("mpx_sa3_pack_weights");                                                                        \
  } while (0)
  SA3_DISPATCH(CALL)
#undef CALL
}

// measurement only: the same launch with s_memtime stamps of workgroup 300, wave 0 at the phase
// boundaries -> probe[0..16] (tools/probes/sa3_phase_probe.py)
MPX_EXPORT int mpx_sa3_chain_probe(const float *x, int ldx, int B, const float *pack, float *out, int ldo, int64_t *probe,
                                   mpx_stream_t stream) {
  using C = sa3::Cfg<272, 512, 512, 1024>;
  MPX_LDS_LIMIT_ONCE((sa3_chain_kernel<272, 512, 512, 1024, true>), C::LDS_BYTES, "mpx_sa3_chain_probe");
  hipLaunchKernelGGL((sa3_chain_kernel<272, 512, 512, 1024, true>), dim3((unsigned)B), dim3(64 * sa3::WV), C::LDS_BYTES,
                     mpx_s(stream), x, ldx, pack, out, ldo, reinterpret_cast<long long *>(probe));
  MPX_LAUNCH_CHECK("mpx_sa3_chain_probe");
}

MPX_EXPORT int mpx_sa3_chain(const float *x, int ldx, int B, int rows, const float *pack, int K1, int c1, int c2, int c3,
                             float *out, int ldo, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && x && pack && out, "mpx_sa3_chain: bad operand");
  MPX_REQUIRE(rows == sa3::ROWS, "mpx_sa3_chain: the module pools exactly %d rows per environment (got %d)", sa3::ROWS, rows);
  MPX_REQUIRE(ldx >= K1 && ldx % 4 == 0 && ldo >= c3, "mpx_sa3_chain: bad leading dimension");
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)pack) & 15) == 0, "mpx_sa3_chain: x and pack must be 16-byte aligned");
  if (B == 0) return 0;
#define CALL(a, b, c, d)                                                                                            \
  do {                                                                                                              \
    using C = sa3::Cfg<a, b, c, d>;                                                                                 \
    MPX_LDS_LIMIT_ONCE((sa3_chain_kernel<a, b, c, d>), C::LDS_BYTES, "mpx_sa3_chain");                              \
    hipLaunchKernelGGL((sa3_chain_kernel<a, b, c, d>), dim3((unsigned)B), dim3(64 * sa3::WV), C::LDS_BYTES,        \
                       mpx_s(stream), x, ldx, pack, out, ldo, (long long *)nullptr);                                \
    MPX_LAUNCH_CHECK("mpx_sa3_chain");                                                                              \
  } while (0)
  SA3_DISPATCH(CALL)
#undef CALL
}
