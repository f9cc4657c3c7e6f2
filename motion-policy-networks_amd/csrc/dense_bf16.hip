// dense_bf16.hip -- split-bf16 ("bf16x3") variant of the dense layers: y = act(x . W^T + b) with every fp32
// product evaluated as x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), the same
// arithmetic as sa_mlp_bf16.hip.  3 MFMAs of 32 cycles per 16 k-values instead of 8 fp32 MFMAs of 64 cycles.
//
// Split operands are kept in the PAIRS form: a row holds, per group of 16 k-values, [hi x 16 | lo x 16] bf16 = 64
// contiguous bytes (hi = bf16(v), lo = bf16(v - hi); K padded to a multiple of 16 with zeros), i.e. [rows, 2 Kp] bf16
// -- the same bytes as the fp32 rows.  A 16-k slab of a row is then ONE 64-byte piece that carries both planes, which
// is what a direct-to-LDS load wants (two separate [rows, Kp] planes, the first form of this file, gave 32-byte pieces
// at 16-k slabs -- slower than splitting on the fly -- or 64 KB stages at 32-k slabs).
//   * weights are split once (mpx_split_bf16);
//   * linear_bf16x3_kernel takes fp32 activations and splits them while staging (any K % 4 == 0; 128 x 128 tile);
//   * linear_bf16x3_pairs_kernel takes activations ALREADY in the pairs form -- written by the epilogue of the layer
//     before, so a value is split once instead of once per 128-column tile that reads it, and nothing is converted on
//     the way in: both operands go global -> LDS by DMA.
// Every kernel accumulates hi*hi, hi*lo, lo*hi per 16 k-values in the same order, so the forms agree bit for bit.
// Opt-in with the other bf16x3 kernels (model.set_precision("bf16x3")).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int HB_BM = 128, HB_BN = 128, HB_BK = 16, HB_LDT = HB_BK + 8;  // bf16 elements per padded LDS row

__device__ __forceinline__ float hb_act(float v, int act) {
  if (act == MPX_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == MPX_ACT_LEAKY) return v >= 0.0f ? v : v * 0.01f;
  return v;
}

// element (row, k) of an operand in the pairs form: hi at pairs_at(...), lo 16 elements behind
__device__ __forceinline__ size_t pairs_at(size_t row, int ld, int k) { return row * (size_t)ld + (size_t)(k >> 4) * 32 + (k & 15); }

// fp32 rows [R, K] (leading dimension ldx) -> pairs [R, ldp >= 2 Kp]
__global__ void __launch_bounds__(256)
    split_bf16_kernel(const float *__restrict__ x, int ldx, int64_t R, int K, int Kp, __bf16 *__restrict__ p, int ldp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * Kp) return;
  const int64_t r = i / Kp;
  const int k = (int)(i - r * Kp);
  const float v = k < K ? x[r * ldx + k] : 0.0f;
  const __bf16 h = (__bf16)v;
  const size_t o = pairs_at((size_t)r, ldp, k);
  p[o] = h;
  p[o + 16] = (__bf16)(v - (float)h);
}

// Store one 32 x 64 block of a wave's results: the accumulators of a 32-row half pass through the wave's own LDS
// staging area (ds_write_b32 in the MFMA layout, back as rows of float4) and leave either as fp32 rows (yp == nullptr)
// or already split, in the pairs form the next layer's kernel stages without touching them (8 bytes per lane and
// plane).
constexpr int HB_LDC = 64 + 4;
__device__ __forceinline__ void hb_store_rows(const float *stage, int lane, int row0, int col0, int M, int N, float *y,
                                              int ldy, bool vec_ok, __bf16 *yp, int ldp, const float *dact_of = nullptr,
                                              int lddact = 0, int dact = 0) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int e = t * 64 + lane;
    const int rr = e >> 4, c4 = (e & 15) * 4;
    const int row = row0 + rr, col = col0 + c4;
    float4 v = *reinterpret_cast<const float4 *>(&stage[rr * HB_LDC + c4]);
    if (row >= M || col >= N) continue;
    if (dact_of != nullptr) {  // (mpx_linear_bf16x3_dact: times act'(output of the layer below), as mpx_act_backward does)
      const float *mk = dact_of + (size_t)row * lddact + col;
      const float m0 = mk[0], m1 = col + 1 < N ? mk[1] : 0.0f, m2 = col + 2 < N ? mk[2] : 0.0f, m3 = col + 3 < N ? mk[3] : 0.0f;
      if (dact == MPX_ACT_RELU) {
        v.x = m0 > 0.0f ? v.x : 0.0f, v.y = m1 > 0.0f ? v.y : 0.0f, v.z = m2 > 0.0f ? v.z : 0.0f, v.w = m3 > 0.0f ? v.w : 0.0f;
      } else if (dact == MPX_ACT_LEAKY) {
        v.x = m0 >= 0.0f ? v.x : 0.01f * v.x, v.y = m1 >= 0.0f ? v.y : 0.01f * v.y;
        v.z = m2 >= 0.0f ? v.z : 0.01f * v.z, v.w = m3 >= 0.0f ? v.w : 0.01f * v.w;
      }
    }
    if (yp != nullptr) {  // (launcher: N is a multiple of 4; the pad columns of the last 16-group are written by the
                          // caller's zero fill or never read: K of the next layer = N)
      const float f[4] = {v.x, v.y, v.z, v.w};
      bf16x4 h, l;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = (__bf16)f[q];
        l[q] = (__bf16)(f[q] - (float)h[q]);
      }
      __bf16 *dst = yp + pairs_at((size_t)row, ldp, col);
      *reinterpret_cast<bf16x4 *>(dst) = h;
      *reinterpret_cast<bf16x4 *>(dst + 16) = l;
      continue;
    }
    float *dst = y + (size_t)row * ldy + col;
    if (vec_ok && col + 3 < N) {
      *reinterpret_cast<float4 *>(dst) = v;
    } else {
      if (col + 0 < N) dst[0] = v.x;
      if (col + 1 < N) dst[1] = v.y;
      if (col + 2 < N) dst[2] = v.z;
      if (col + 3 < N) dst[3] = v.w;
    }
  }
}

template <bool POOL>
__global__ void __launch_bounds__(256)
    linear_bf16x3_kernel(const float *__restrict__ x, int ldx, const __bf16 *__restrict__ w, int ldw,
                         const float *__restrict__ bias, int M, int N, int K, int Kp, int act, float *__restrict__ y,
                         int ldy, __bf16 *__restrict__ yp, int ldp, const float *__restrict__ dact_of, int lddact, int dact,
                         const int32_t *__restrict__ seg, int seg_row0) {
  // [stage][plane: A_hi, A_lo, B_hi, B_lo][128 rows x LDT]
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 4 * HB_BM * HB_LDT];
  auto plane = [&](int stage, int p) { return smem + ((stage * 4 + p) * HB_BM) * HB_LDT; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  if ((gridDim.y & 7) == 0) {  // all N-tiles of a 128-row block on one XCD (see dense.hip)
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7, slot = lin >> 3;
    by = (int)((slot / gridDim.x) * 8 + xcd);
    bx = (int)(slot % gridDim.x);
  }
  const int m0 = by * HB_BM, n0 = bx * HB_BN;

  // staging maps: x (fp32) 2 float4 per thread; w (pairs: the slab of a row is [hi x 16 | lo x 16]) two 16-byte loads
  const int xr = tid >> 2, xc = (tid & 3) * 4;
  const int wr = tid >> 1, wc = (tid & 1) * 8;
  float4 px[2];
  bf16x8 pwh, pwl;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = m0 + xr + 64 * i, kk = k0 + xc;
      px[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < M && kk < K) px[i] = *reinterpret_cast<const float4 *>(x + (size_t)r * ldx + kk);
    }
    const int n = n0 + wr;
    pwh = pwl = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N) {
      const __bf16 *src = w + (size_t)n * ldw + 2 * k0 + wc;  // (k0 is a multiple of 16: group k0 / 16 starts at 2 k0)
      pwh = *reinterpret_cast<const bf16x8 *>(src);
      pwl = *reinterpret_cast<const bf16x8 *>(src + 16);
    }
  };
  auto sstore = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v[4] = {px[i].x, px[i].y, px[i].z, px[i].w};
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = (__bf16)v[e];
        l[e] = (__bf16)(v[e] - (float)h[e]);
      }
      const int r = xr + 64 * i;
      *reinterpret_cast<bf16x4 *>(plane(stage, 0) + r * HB_LDT + xc) = h;
      *reinterpret_cast<bf16x4 *>(plane(stage, 1) + r * HB_LDT + xc) = l;
    }
    *reinterpret_cast<bf16x8 *>(plane(stage, 2) + wr * HB_LDT + wc) = pwh;
    *reinterpret_cast<bf16x8 *>(plane(stage, 3) + wr * HB_LDT + wc) = pwl;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  const int nk = Kp / HB_BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kb = 0; kb < nk; ++kb) {
    const int st = kb & 1;
    if (kb + 1 < nk) gload((kb + 1) * HB_BK);
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // operand k = 8*half + e: eight consecutive bf16 of the row
      const int ra = (wm * 64 + t * 32 + l31) * HB_LDT + 8 * half, rb = (wn * 64 + t * 32 + l31) * HB_LDT + 8 * half;
      ah[t] = *reinterpret_cast<const bf16x8 *>(plane(st, 0) + ra);
      al[t] = *reinterpret_cast<const bf16x8 *>(plane(st, 1) + ra);
      bh[t] = *reinterpret_cast<const bf16x8 *>(plane(st, 2) + rb);
      bl[t] = *reinterpret_cast<const bf16x8 *>(plane(st, 3) + rb);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (kb + 1 < nk) sstore(st ^ 1);
    __syncthreads();
  }

  // epilogue: C[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*half (same as the fp32 kernel)
  if (POOL && seg != nullptr) {  // max over each SEGMENT of rows (training; y = the 64-bit keys, ldy in keys: common.h)
    float bvj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      bvj[j] = (bias && col < N) ? bias[col] : 0.0f;
    }
    mpx_segpool_tile([&](int i, int j, int r) __attribute__((always_inline)) { return hb_act(acc[i][j][r] + bvj[j], act); },
                     m0 + wm * 64, n0 + wn * 64, M, N, half, l31, seg, seg_row0, reinterpret_cast<unsigned long long *>(y), ldy);
  } else if (POOL) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.0f;
      float m = 0.0f;  // post-ReLU values are >= 0
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < M) m = fmaxf(m, hb_act(acc[i][j][r] + bv, act));
        }
      m = mpx_max_across_halves(m);
      if (half == 0) atomicMax(reinterpret_cast<int *>(y + (size_t)by * ldy + col), __float_as_int(m));
    }
  } else {
    constexpr int LDC = HB_LDC;
    float *stage = reinterpret_cast<float *>(smem) + wave * (32 * LDC);  // 4 x 8.5 KB inside the 48 KB of operand planes
    static_assert(4 * 32 * LDC * 4 <= 2 * 4 * HB_BM * HB_LDT * 2, "staging must fit the operand planes");
    const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * half) * LDC + j * 32 + l31] = hb_act(acc[i][j][r] + bv, act);
      }
      hb_store_rows(stage, lane, m0 + wm * 64 + i * 32, n0 + wn * 64, M, N, y, ldy, vec_ok, yp, ldp, dact_of, lddact, dact);
    }
  }
}

// ---- pairs-input form ------------------------------------------------------------------------------------------------
// 256 x 128 tile, four waves as 2 x 2 with a 128 x 64 wave tile (12 fragment reads per 24 MFMAs), slabs of 16 k (one
// 64-byte piece per row and operand), three-stage ring (24 KB a stage: two workgroups per CU) with counted vmcnt: each
// wave waits only for its own loads of the NEXT slab before the one barrier of a slab, the slab after stays in flight
// across it (`__syncthreads()` would drain it: its fence counts an LDS-DMA as a pending LDS write; the raw s_barrier is
// preceded by an explicit lgkmcnt(0) for the fragment reads, after which the stage may be refilled).  A wave's DMA load
// writes 16 rows x 64 B of consecutive LDS; the 16-byte chunk c of row r sits at physical chunk c ^ ((r >> 2) & 3) (the
// lane picks which global chunk it fetches, the fragment reads apply the same xor: every quarter-wave ds_read_b128
// touches all 64 banks once -- the layout of dense.hip's DMA kernel).  Rows past M / N read as zeros through the
// buffer descriptor's range check.
// OUT: 0 fp32 rows, 1 pairs, 2 max over each 128-row group (= one wave's rows; post-ReLU) as fp32, 3 the same as pairs.
// Measured at the group-all shapes of 8192 environments (M = 1 M rows): 512 -> 1024 + pooling 2.73 ms = 1.21 PFLOP/s of
// bf16 MFMA work (fp32-row kernel 3.93 ms; separate-planes kernel with 32-k slabs, 128 x 128 tiles 3.32 ms),
// 512 -> 512 pairs to pairs 1.78 ms (2.43; 2.17).
constexpr int Y_BM = 256, Y_BN = 128, Y_STAGES = 3, Y_ROWB = 64;
constexpr int Y_STAGE = (Y_BM + Y_BN) * Y_ROWB, Y_LDS = Y_STAGES * Y_STAGE;
template <int OUT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
    linear_bf16x3_pairs_kernel(const __bf16 *__restrict__ a, int lda, const __bf16 *__restrict__ w, int ldw, int K,
                               const float *__restrict__ bias, int M, int N, int act, float *__restrict__ y, int ldy,
                               __bf16 *__restrict__ yp, int ldp) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ysm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the DMA loads' LDS addresses stay on the scalar unit)
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  if ((gridDim.y & 7) == 0) {  // all N-tiles of a row block on one XCD (see dense.hip)
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7, slot = lin >> 3;
    by = (int)((slot / gridDim.x) * 8 + xcd);
    bx = (int)(slot % gridDim.x);
  }
  const int m0 = by * Y_BM, n0 = bx * Y_BN;
  typedef __attribute__((address_space(3))) void lds_void;
  const __amdgpu_buffer_rsrc_t r_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(a), 0, (int)(uint32_t)((int64_t)M * lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(w), 0, (int)(uint32_t)((int64_t)N * ldw * 2), 0x00020000);
  // wave `wave` brings rows [64 wave, +64) of A (4 loads of 16 rows) and rows [32 wave, +32) of B (2 loads):
  // lane -> (row = lane / 4, physical chunk = lane % 4), fetching logical chunk (lane % 4) ^ (lane / 16)
  int avoff[4], wvoff[2];
  {
    const int c16 = (((lane & 3) ^ (lane >> 4)) & 3) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 * wave + 16 * i + (lane >> 2);
      avoff[i] = m0 + r < M ? (int)((uint32_t)(m0 + r) * (uint32_t)lda * 2u + (uint32_t)c16) : (int)0xFFFFFFF0u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * wave + 16 * i + (lane >> 2);
      wvoff[i] = n0 + r < N ? (int)((uint32_t)(n0 + r) * (uint32_t)ldw * 2u + (uint32_t)c16) : (int)0xFFFFFFF0u;
    }
  }
  auto dma = [&](int kb, int stage) __attribute__((always_inline)) {
    unsigned char *sa = ysm + stage * Y_STAGE + (64 * wave) * Y_ROWB;
    unsigned char *sb = ysm + stage * Y_STAGE + Y_BM * Y_ROWB + (32 * wave) * Y_ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (lds_void *)(sa + 16 * i * Y_ROWB), 16, avoff[i], kb * 64, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void *)(sb + 16 * i * Y_ROWB), 16, wvoff[i], kb * 64, 0, 0);
  };
  // fragment addresses, once per kernel: row r of A (isb = 0) or B (1), logical chunk c = 2 * plane + half sits at
  // physical chunk c ^ (r >> 2) -- the lo plane is the hi address with bit 5 flipped, so both get a register; the stage is
  // an immediate of the read (round 5: the slab loop carried 22 address VALU + 6 v_readfirstlane per 24 MFMAs and read
  // its fragments only after the slab's barrier -- an LDS round trip in front of every slab's first MFMA)
  int fa_hi[4], fa_lo[4], fb_hi[2], fb_lo[2];
  auto faddr = [&](int isb, int r, int c) __attribute__((always_inline)) {
    return isb * (Y_BM * Y_ROWB) + r * Y_ROWB + (((c ^ (r >> 2)) & 3) << 4);
  };
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    fa_hi[t] = faddr(0, wm * 128 + t * 32 + l31, half);
    fa_lo[t] = faddr(0, wm * 128 + t * 32 + l31, 2 + half);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    fb_hi[t] = faddr(1, wn * 64 + t * 32 + l31, half);
    fb_lo[t] = faddr(1, wn * 64 + t * 32 + l31, 2 + half);
  }
  auto frag_at = [&](int stage, int off) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8 *>(ysm + off + stage * Y_STAGE);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int nk = K / 16;
  constexpr int LPS = 6;                       // DMA loads per wave and slab (dma(): 4 of A + 2 of B)
  static_assert(Y_STAGES == 3, "the slab loop is written for a three-stage ring (st1 / st2 arithmetic, the 6-way unroll, kb % 3)");
  static_assert(LPS == Y_BM / 64 + Y_BN / 64 && Y_BM == 256 && Y_BN == 128,
                "WAIT_ONE counts the loads one dma() issues per wave: 16 rows each, a wave brings 64 rows of A and 32 of B");
  constexpr int WAIT_ONE = 0x0070 | LPS;       // vmcnt(6), lgkmcnt(0): everything but the youngest slab has landed
  constexpr int WAIT_ALL = 0x0070;             // vmcnt(0), lgkmcnt(0)
  // Software pipeline of a slab (24 MFMAs per wave): the B fragments and A tiles 0-1 of slab k+1 are read into a second
  // register set in the MIDDLE of slab k (behind the one barrier of the slab, which also covers the DMA of slab k+1 --
  // each wave waits for its own loads first), A tiles 2-3 of slab k at its top: every LDS read has half a slab of MFMAs
  // between issue and use.  Three LDS stages: slab k+2 is requested at the top of slab k into the stage slab k-1 was read
  // from -- all waves finished those reads before the barrier in the middle of slab k-1 (lgkmcnt(0) in front of it).
  // (two register sets for the prefetched fragments, alternating by slab: the loop is unrolled over 3 stages x 2 sets)
  bf16x8 bh[2][2], bl[2][2], a01h[2][2], a01l[2][2], a23h[2], a23l[2];
  auto read_front = [&](int stage, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bh[set][t] = frag_at(stage, fb_hi[t]);
      bl[set][t] = frag_at(stage, fb_lo[t]);
      a01h[set][t] = frag_at(stage, fa_hi[t]);
      a01l[set][t] = frag_at(stage, fa_lo[t]);
    }
  };
  auto mfma3 = [&](f32x16 &c, const bf16x8 &xh, const bf16x8 &xl, const bf16x8 &wh, const bf16x8 &wl) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, c, 0, 0, 0);
  };
  if (nk > 0) dma(0, 0);
  if (nk > 1) dma(1, 1);
  if (nk > 1) __builtin_amdgcn_s_waitcnt(WAIT_ONE); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
  __builtin_amdgcn_s_barrier();
  if (nk > 0) read_front(0, 0);
  auto slab = [&](int kb, int stage, int set) __attribute__((always_inline)) {  // stage = kb % 3, set = kb % 2 (compile-time)
    const int st1 = stage == 2 ? 0 : stage + 1, st2 = stage == 0 ? 2 : stage - 1;  // stages of slabs kb + 1, kb + 2
    if (kb + 2 < nk) dma(kb + 2, st2);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a23h[t] = frag_at(stage, fa_hi[2 + t]);
      a23l[t] = frag_at(stage, fa_lo[2 + t]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mfma3(acc[i][j], a01h[set][i], a01l[set][i], bh[set][j], bl[set][j]);
    if (kb + 1 < nk) {
      // own loads of slab kb + 1 landed, own reads done
      if (kb + 2 < nk) __builtin_amdgcn_s_waitcnt(WAIT_ONE); else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
      __builtin_amdgcn_s_barrier();
      read_front(st1, set ^ 1);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mfma3(acc[2 + i][j], a23h[i], a23l[i], bh[set][j], bl[set][j]);
  };
  for (int kb = 0; kb < nk; kb += 6) {
    slab(kb, 0, 0);
    if (kb + 1 < nk) slab(kb + 1, 1, 1);
    if (kb + 2 < nk) slab(kb + 2, 2, 0);
    if (kb + 3 < nk) slab(kb + 3, 0, 1);
    if (kb + 4 < nk) slab(kb + 4, 1, 0);
    if (kb + 5 < nk) slab(kb + 5, 2, 1);
  }
  __builtin_amdgcn_s_waitcnt(WAIT_ALL);
  __builtin_amdgcn_s_barrier();  // (the epilogue reuses the ring as its staging area)
  if (OUT >= 2) {  // max over the wave's 128 rows (one pooled group), plain stores: no other wave owns these columns
    const int grp = by * 2 + wm;
    if ((int64_t)grp * 128 < M) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
        float m = 0.0f;  // post-ReLU values are >= 0
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, hb_act(acc[i][j][r] + bv, act));
        m = mpx_max_across_halves(m);
        if (half == 0 && col < N) {
          if (OUT == 3) {  // the pooled row leaves in the pairs form (the next dense layer's operand)
            const __bf16 h = (__bf16)m;
            const size_t o = pairs_at((size_t)grp, ldp, col);
            yp[o] = h;
            yp[o + 16] = (__bf16)(m - (float)h);
          } else {
            y[(size_t)grp * ldy + col] = m;
          }
        }
      }
    }
  } else {
    float *stage = reinterpret_cast<float *>(ysm) + wave * (32 * HB_LDC);  // (the ring is free after the last barrier)
    const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * half) * HB_LDC + j * 32 + l31] = hb_act(acc[i][j][r] + bv, act);
      }
      hb_store_rows(stage, lane, m0 + wm * 128 + i * 32, n0 + wn * 64, M, N, y, ldy, vec_ok, OUT == 1 ? yp : nullptr, ldp);
    }
  }
}

// ---- host entry points --------------------------------------------------------------------------------------------
static int hb_check_w(const char *name, const void *w, int N, int K) {
  MPX_REQUIRE(N >= 1 && K >= 1 && w, "%s: bad weight operand", name);
  MPX_REQUIRE(((uintptr_t)w & 15) == 0, "%s: the weight pairs must be 16-byte aligned", name);
  return 0;
}
static int hb_kp(int K) { return (K + 15) / 16 * 16; }

MPX_EXPORT int mpx_split_bf16(const float *x, int ldx, int64_t R, int K, void *pairs, int ldp, mpx_stream_t stream) {
  MPX_REQUIRE(R >= 0 && K >= 1 && x && pairs && ldx >= K, "mpx_split_bf16: bad argument");
  MPX_REQUIRE(ldp >= 2 * hb_kp(K), "mpx_split_bf16: ldp must be at least 2 * roundup(K, 16) = %d", 2 * hb_kp(K));
  if (R == 0) return 0;
  const int64_t n = R * hb_kp(K);
  MPX_REQUIRE(n / 256 < ((int64_t)1 << 31) - 1, "mpx_split_bf16: too large");
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mpx_s(stream), x, ldx, R, K, hb_kp(K),
                     reinterpret_cast<__bf16 *>(pairs), ldp);
  MPX_LAUNCH_CHECK("mpx_split_bf16");
}

static int hb_check(const char *name, const float *x, int ldx, const void *w, int M, int N, int K, int ldy) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "%s: bad size", name);
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "%s: K and ldx must be multiples of 4 (got %d, %d)", name, K, ldx);
  MPX_REQUIRE(((uintptr_t)x & 15) == 0, "%s: operands must be 16-byte aligned", name);
  MPX_REQUIRE(ldx >= K && ldy >= N, "%s: leading dimension too small", name);
  return hb_check_w(name, w, N, K);
}
static int hb_check_pairs_out(const char *name, const void *yp, int N, int ldp) {
  MPX_REQUIRE(yp, "%s: output pairs missing", name);
  MPX_REQUIRE(N % 4 == 0 && ldp % 4 == 0 && ldp >= 2 * hb_kp(N), "%s: N and ldp must be multiples of 4, ldp >= %d (got %d, %d)",
              name, 2 * hb_kp(N), N, ldp);
  MPX_REQUIRE(((uintptr_t)yp & 7) == 0, "%s: output pairs must be 8-byte aligned", name);
  return 0;
}

MPX_EXPORT int mpx_linear_bf16x3(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N, int K, int act,
                                 float *y, int ldy, mpx_stream_t stream) {
  if (hb_check("mpx_linear_bf16x3", x, ldx, w_pairs, M, N, K, ldy)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_bf16x3: unknown activation %d", act);
  if (M == 0) return 0;
  if (const int64_t slab = mpx_row_slab(HB_BM, 0); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_bf16x3(x + m0 * ldx, ldx, w_pairs, bias, (int)(M - m0 < slab ? M - m0 : slab), N, K, act,
                                     y + m0 * ldy, ldy, stream))
        return rc;
    return 0;
  }
  hipLaunchKernelGGL((linear_bf16x3_kernel<false>), dim3(cdiv(N, HB_BN), cdiv(M, HB_BM)), dim3(256), 0, mpx_s(stream), x,
                     ldx, reinterpret_cast<const __bf16 *>(w_pairs), 2 * hb_kp(K), bias, M, N, K, hb_kp(K), act, y, ldy,
                     (__bf16 *)nullptr, 0, (const float *)nullptr, 0, 0, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3");
}

MPX_EXPORT int mpx_linear_bf16x3_to_pairs(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N, int K,
                                          int act, void *y_pairs, int ldp, mpx_stream_t stream) {
  if (hb_check("mpx_linear_bf16x3_to_pairs", x, ldx, w_pairs, M, N, K, N)) return 1;
  if (hb_check_pairs_out("mpx_linear_bf16x3_to_pairs", y_pairs, N, ldp)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_bf16x3_to_pairs: unknown activation %d", act);
  if (M == 0) return 0;
  if (const int64_t slab = mpx_row_slab(HB_BM, 0); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_bf16x3_to_pairs(x + m0 * ldx, ldx, w_pairs, bias, (int)(M - m0 < slab ? M - m0 : slab), N, K, act,
                                              static_cast<__bf16 *>(y_pairs) + m0 * ldp, ldp, stream))
        return rc;
    return 0;
  }
  hipLaunchKernelGGL((linear_bf16x3_kernel<false>), dim3(cdiv(N, HB_BN), cdiv(M, HB_BM)), dim3(256), 0, mpx_s(stream), x,
                     ldx, reinterpret_cast<const __bf16 *>(w_pairs), 2 * hb_kp(K), bias, M, N, K, hb_kp(K), act,
                     (float *)nullptr, 0, reinterpret_cast<__bf16 *>(y_pairs), ldp, (const float *)nullptr, 0, 0, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3_to_pairs");
}

MPX_EXPORT int mpx_linear_rowmax_bf16x3(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N, int K,
                                        int rows, float *y, int ldy, mpx_stream_t stream) {
  if (hb_check("mpx_linear_rowmax_bf16x3", x, ldx, w_pairs, M, N, K, ldy)) return 1;
  MPX_REQUIRE(rows == HB_BM && M % HB_BM == 0, "mpx_linear_rowmax_bf16x3: pooled groups must be exactly %d rows", HB_BM);
  if (M == 0) return 0;
  if (const int64_t slab = mpx_row_slab(HB_BM, 0); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_rowmax_bf16x3(x + m0 * ldx, ldx, w_pairs, bias, (int)(M - m0 < slab ? M - m0 : slab), N, K, rows,
                                            y + (m0 / HB_BM) * ldy, ldy, stream))
        return rc;
    return 0;
  }
  hipError_t e = hipMemset2DAsync(y, (size_t)ldy * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)(M / HB_BM),
                                  mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_linear_rowmax_bf16x3: memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((linear_bf16x3_kernel<true>), dim3(cdiv(N, HB_BN), M / HB_BM), dim3(256), 0, mpx_s(stream), x, ldx,
                     reinterpret_cast<const __bf16 *>(w_pairs), 2 * hb_kp(K), bias, M, N, K, hb_kp(K), (int)MPX_ACT_RELU, y,
                     ldy, (__bf16 *)nullptr, 0, (const float *)nullptr, 0, 0, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear_rowmax_bf16x3");
}

// the grouped MLPs' last layer + activation + max over each query's rows (dense.hip: mpx_linear_segmax) in split bf16
int mpx_segmax_unpack_launch(const unsigned long long *keys, int64_t Q, int N, float *pooled, int ldp, int64_t *arg,
                             hipStream_t stream);  // dense.hip
int mpx_segmax_check(const char *name, int M, const int32_t *seg, int64_t Q, int N, const void *keys, const float *pooled,
                     int ldp, const int64_t *arg);
MPX_EXPORT int mpx_linear_segmax_bf16x3(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N, int K,
                                        int act, const int32_t *seg, int64_t Q, void *keys, float *pooled, int ldp,
                                        int64_t *arg, mpx_stream_t stream) {
  if (hb_check("mpx_linear_segmax_bf16x3", x, ldx, w_pairs, M, N, K, N)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_segmax_bf16x3: unknown activation %d", act);
  if (mpx_segmax_check("mpx_linear_segmax_bf16x3", M, seg, Q, N, keys, pooled, ldp, arg)) return 1;
  hipError_t e = hipMemsetAsync(keys, 0, (size_t)Q * N * 8, mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_linear_segmax_bf16x3: memset failed: %s", hipGetErrorString(e));
  const int64_t slab = mpx_row_slab(HB_BM, 0);
  for (int64_t m0 = 0; m0 < M; m0 += slab) {
    const int m = (int)(M - m0 < slab ? M - m0 : slab);
    hipLaunchKernelGGL((linear_bf16x3_kernel<true>), dim3(cdiv(N, HB_BN), cdiv(m, HB_BM)), dim3(256), 0, mpx_s(stream),
                       x + m0 * ldx, ldx, reinterpret_cast<const __bf16 *>(w_pairs), 2 * hb_kp(K), bias, m, N, K, hb_kp(K), act,
                       reinterpret_cast<float *>(keys), N, (__bf16 *)nullptr, 0, (const float *)nullptr, 0, 0, seg + m0, (int)m0);
  }
  mpx_segmax_unpack_launch(static_cast<const unsigned long long *>(keys), Q, N, pooled, ldp, arg, mpx_s(stream));
  MPX_LAUNCH_CHECK("mpx_linear_segmax_bf16x3");
}

static int pb_check(const char *name, const void *a, int lda, const void *w, int M, int N, int K) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 16, "%s: bad size", name);
  MPX_REQUIRE(a, "%s: activation pairs missing", name);
  MPX_REQUIRE(K % 16 == 0 && lda % 8 == 0 && lda >= 2 * K, "%s: K must be a multiple of 16, lda of 8 and >= 2 K (got %d, %d)",
              name, K, lda);
  MPX_REQUIRE(((uintptr_t)a & 15) == 0, "%s: the activation pairs must be 16-byte aligned", name);
  MPX_REQUIRE((int64_t)N * K * 4 < ((int64_t)1 << 32) - 16, "%s: the weight pairs must stay under 4 GB", name);
  // (rows: the entry points walk them in slabs of mpx_row_slab(Y_BM, lda * 2): activation pairs under 4 GB per launch)
  return hb_check_w(name, w, N, K);
}
#define PB_LAUNCH(OUT, grid, s, ...)                                                                       \
  do {                                                                                                      \
    MPX_LDS_LIMIT_ONCE(linear_bf16x3_pairs_kernel<OUT>, Y_LDS, "mpx_linear_bf16x3_pairs");                  \
    hipLaunchKernelGGL((linear_bf16x3_pairs_kernel<OUT>), grid, dim3(256), Y_LDS, s, __VA_ARGS__);          \
  } while (0)

MPX_EXPORT int mpx_linear_bf16x3_pairs(const void *a_pairs, int lda, const void *w_pairs, const float *bias, int M, int N,
                                       int K, int act, float *y, int ldy, void *y_pairs, int ldp, mpx_stream_t stream) {
  if (pb_check("mpx_linear_bf16x3_pairs", a_pairs, lda, w_pairs, M, N, K)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_bf16x3_pairs: unknown activation %d", act);
  MPX_REQUIRE((y != nullptr) != (y_pairs != nullptr), "mpx_linear_bf16x3_pairs: pass either y (fp32 rows) or y_pairs");
  if (y) {
    MPX_REQUIRE(ldy >= N, "mpx_linear_bf16x3_pairs: leading dimension too small");
  } else if (hb_check_pairs_out("mpx_linear_bf16x3_pairs", y_pairs, N, ldp)) {
    return 1;
  }
  if (M == 0) return 0;
  if (const int64_t slab = mpx_row_slab(Y_BM, (int64_t)lda * 2); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_bf16x3_pairs(static_cast<const __bf16 *>(a_pairs) + m0 * lda, lda, w_pairs, bias,
                                           (int)(M - m0 < slab ? M - m0 : slab), N, K, act, y ? y + m0 * ldy : nullptr, ldy,
                                           y_pairs ? static_cast<__bf16 *>(y_pairs) + m0 * ldp : nullptr, ldp, stream))
        return rc;
    return 0;
  }
  const dim3 grid(cdiv(N, Y_BN), cdiv(M, Y_BM));
  const __bf16 *ap = reinterpret_cast<const __bf16 *>(a_pairs), *wp = reinterpret_cast<const __bf16 *>(w_pairs);
  if (y)
    PB_LAUNCH(0, grid, mpx_s(stream), ap, lda, wp, 2 * K, K, bias, M, N, act, y, ldy, (__bf16 *)nullptr, 0);
  else
    PB_LAUNCH(1, grid, mpx_s(stream), ap, lda, wp, 2 * K, K, bias, M, N, act, (float *)nullptr, 0,
              reinterpret_cast<__bf16 *>(y_pairs), ldp);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3_pairs");
}

MPX_EXPORT int mpx_linear_rowmax_bf16x3_pairs(const void *a_pairs, int lda, const void *w_pairs, const float *bias, int M,
                                              int N, int K, int rows, float *y, int ldy, void *y_pairs, int ldp,
                                              mpx_stream_t stream) {
  if (pb_check("mpx_linear_rowmax_bf16x3_pairs", a_pairs, lda, w_pairs, M, N, K)) return 1;
  MPX_REQUIRE(rows == 128 && M % 128 == 0, "mpx_linear_rowmax_bf16x3_pairs: pooled groups must be exactly 128 rows");
  MPX_REQUIRE((y != nullptr) != (y_pairs != nullptr), "mpx_linear_rowmax_bf16x3_pairs: pass either y (fp32 rows) or y_pairs");
  if (y) {
    MPX_REQUIRE(ldy >= N, "mpx_linear_rowmax_bf16x3_pairs: leading dimension too small");
  } else {
    MPX_REQUIRE(ldp >= 2 * hb_kp(N) && ((uintptr_t)y_pairs & 1) == 0, "mpx_linear_rowmax_bf16x3_pairs: bad output pairs");
  }
  if (M == 0) return 0;
  if (const int64_t slab = mpx_row_slab(Y_BM, (int64_t)lda * 2); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_rowmax_bf16x3_pairs(static_cast<const __bf16 *>(a_pairs) + m0 * lda, lda, w_pairs, bias,
                                                  (int)(M - m0 < slab ? M - m0 : slab), N, K, rows,
                                                  y ? y + (m0 / 128) * ldy : nullptr, ldy,
                                                  y_pairs ? static_cast<__bf16 *>(y_pairs) + (m0 / 128) * ldp : nullptr, ldp, stream))
        return rc;
    return 0;
  }
  const dim3 grid(cdiv(N, Y_BN), cdiv(M, Y_BM));
  const __bf16 *ap = reinterpret_cast<const __bf16 *>(a_pairs), *wp = reinterpret_cast<const __bf16 *>(w_pairs);
  if (y)
    PB_LAUNCH(2, grid, mpx_s(stream), ap, lda, wp, 2 * K, K, bias, M, N, (int)MPX_ACT_RELU, y, ldy, (__bf16 *)nullptr, 0);
  else
    PB_LAUNCH(3, grid, mpx_s(stream), ap, lda, wp, 2 * K, K, bias, M, N, (int)MPX_ACT_RELU, (float *)nullptr, 0,
              reinterpret_cast<__bf16 *>(y_pairs), ldp);
  MPX_LAUNCH_CHECK("mpx_linear_rowmax_bf16x3_pairs");
}

// ---- training (row N1) in the split-bf16 arithmetic ("AMP" of the reference, run_training.py:112 precision=16, with fp32
// master weights and fp32 accumulation) ------------------------------------------------------------------------------
// dX with the elementwise backward of the layer below in the epilogue (dense.hip: mpx_linear_dact): y = (x . w^T) * act'(dact_of)
MPX_EXPORT int mpx_linear_bf16x3_dact(const float *x, int ldx, const void *w_pairs, int M, int N, int K, const float *dact_of,
                                      int lddact, int dact, float *y, int ldy, mpx_stream_t stream) {
  if (hb_check("mpx_linear_bf16x3_dact", x, ldx, w_pairs, M, N, K, ldy)) return 1;
  MPX_REQUIRE(dact == MPX_ACT_NONE || (dact_of != nullptr && lddact >= N && (dact == MPX_ACT_RELU || dact == MPX_ACT_LEAKY)),
              "mpx_linear_bf16x3_dact: the activation's output rows are missing or too short, or the activation is unknown");
  if (M == 0) return 0;
  if (dact == MPX_ACT_NONE) dact_of = nullptr;
  if (const int64_t slab = mpx_row_slab(HB_BM, 0); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_bf16x3_dact(x + m0 * ldx, ldx, w_pairs, (int)(M - m0 < slab ? M - m0 : slab), N, K,
                                          dact_of ? dact_of + m0 * lddact : nullptr, lddact, dact, y + m0 * ldy, ldy, stream))
        return rc;
    return 0;
  }
  hipLaunchKernelGGL((linear_bf16x3_kernel<false>), dim3(cdiv(N, HB_BN), cdiv(M, HB_BM)), dim3(256), 0, mpx_s(stream), x,
                     ldx, reinterpret_cast<const __bf16 *>(w_pairs), 2 * hb_kp(K), (const float *)nullptr, M, N, K, hb_kp(K),
                     (int)MPX_ACT_NONE, y, ldy, (__bf16 *)nullptr, 0, dact_of, lddact, dact, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3_dact");
}

// dW [N, K] = dY^T . X, db = column sums of dY: dense_grad.hip's linear_wgrad_kernel<128> with every product as three
// bf16 MFMAs.  A slab of 16 batch rows is ONE K16 step of v_mfma_f32_32x32x16_bf16; both operands are split hi / lo while
// they are staged transposed into LDS (a thread holds two adjacent batch rows of four columns: each column's pair goes out
// as one packed 4-byte store per plane; rows of 16 bf16 padded to 48 bytes: the 16-byte fragment reads of a quarter wave
// touch every bank once).  Same split reduction over the rows and the same partial layout as the fp32 kernel.
constexpr int WB_T = 128, WB_LDT = 24;  // tile; bf16 elements per padded LDS row (16 used)
__device__ __forceinline__ void wb_split2(float v0, float v1, unsigned &hi, unsigned &lo) {
  const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
  const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
  hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}
__global__ void __launch_bounds__(256)
    linear_wgrad_bf16x3_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ldx, int M, int N,
                               int K, int rows_per_split, float *__restrict__ partial, int with_bias) {
  // [buffer][plane: A_hi, A_lo, B_hi, B_lo][128 rows x WB_LDT]
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 4 * WB_T * WB_LDT];
  auto plane = [&](int buf, int p) { return smem + ((buf * 4 + p) * WB_T) * WB_LDT; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int k0 = blockIdx.x * WB_T, n0 = blockIdx.y * WB_T;  // tile of dW [N, K]: rows n, columns k
  const int mb = blockIdx.z * rows_per_split, me = min(M, mb + rows_per_split);
  // staging: thread -> batch rows (2 rp, 2 rp + 1) of the slab, columns sc .. sc + 3 of the tile
  const int rp = tid >> 5, sc = (tid & 31) * 4;
  float4 pa[2], pb[2];
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + 2 * rp + i;
      pa[i] = pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < me) {
        if (n0 + sc < N) pa[i] = *reinterpret_cast<const float4 *>(dy + (size_t)m * lddy + n0 + sc);
        if (k0 + sc < K) pb[i] = *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + k0 + sc);
      }
    }
  };
  // db = column sums of dY, from the fp32 values as they pass through the staging registers (a plain sum: no matrix
  // cores, no reason to take the 2^-17-accurate split): a thread adds its two batch rows of four columns per slab, the
  // eight row groups of a column are added in a fixed order at the end
  const bool do_bias = with_bias && blockIdx.x == 0;
  float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto sstore = [&](int buf) {  // transposed: LDS row = output index (n or k), elements 2 rp, 2 rp + 1 of the row
    const float a0[4] = {pa[0].x, pa[0].y, pa[0].z, pa[0].w}, a1[4] = {pa[1].x, pa[1].y, pa[1].z, pa[1].w};
    const float b0[4] = {pb[0].x, pb[0].y, pb[0].z, pb[0].w}, b1[4] = {pb[1].x, pb[1].y, pb[1].z, pb[1].w};
    if (do_bias) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bsum[c] += a0[c] + a1[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned hi, lo;
      wb_split2(a0[c], a1[c], hi, lo);
      *reinterpret_cast<unsigned *>(plane(buf, 0) + (sc + c) * WB_LDT + 2 * rp) = hi;
      *reinterpret_cast<unsigned *>(plane(buf, 1) + (sc + c) * WB_LDT + 2 * rp) = lo;
      wb_split2(b0[c], b1[c], hi, lo);
      *reinterpret_cast<unsigned *>(plane(buf, 2) + (sc + c) * WB_LDT + 2 * rp) = hi;
      *reinterpret_cast<unsigned *>(plane(buf, 3) + (sc + c) * WB_LDT + 2 * rp) = lo;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int nslab = (me - mb + 15) / 16;
  if (nslab > 0) {
    gload(mb);
    sstore(0);
  }
  __syncthreads();
  for (int kb = 0; kb < nslab; ++kb) {
    const int buf = kb & 1;
    if (kb + 1 < nslab) gload(mb + (kb + 1) * 16);
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // operand element e of lane-half h: batch row 8 h + e of the slab
      const int ra = (wm * 64 + t * 32 + l31) * WB_LDT + 8 * half, rb = (wn * 64 + t * 32 + l31) * WB_LDT + 8 * half;
      ah[t] = *reinterpret_cast<const bf16x8 *>(plane(buf, 0) + ra);
      al[t] = *reinterpret_cast<const bf16x8 *>(plane(buf, 1) + ra);
      bh[t] = *reinterpret_cast<const bf16x8 *>(plane(buf, 2) + rb);
      bl[t] = *reinterpret_cast<const bf16x8 *>(plane(buf, 3) + rb);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (kb + 1 < nslab) sstore(buf ^ 1);
    __syncthreads();
  }
  // per split: [N*K weight partials | N bias partials]; C[row][col]: col = lane&31 (k), row = (r&3) + 8*(r>>2) + 4*half (n)
  float *dst = partial + (size_t)blockIdx.z * ((size_t)N * K + N);
  if (do_bias) {  // (the operand planes are free: every wave passed the slab loop's last barrier)
    float *red = reinterpret_cast<float *>(smem);  // [8 row groups][128 columns]
    static_assert(8 * WB_T * 4 <= sizeof(smem), "the bias reduction fits the operand planes");
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rp * WB_T + sc + c] = bsum[c];
    __syncthreads();
    if (tid < WB_T && n0 + tid < N) {
      float t = red[tid];
#pragma unroll
      for (int g = 1; g < 8; ++g) t += red[g * WB_T + tid];
      dst[(size_t)N * K + n0 + tid] = t;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < N && k < K) dst[(size_t)n * K + k] = acc[i][j][r];
      }
    }
}
// (host helper for dense_grad.hip's mpx_linear_wgrad_bf16x3: the split reduction and its partial layout live there)
void mpx_wgrad_bf16x3_launch(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, int rows_per_split,
                             int S, float *partial, int with_bias, hipStream_t stream) {
  hipLaunchKernelGGL(linear_wgrad_bf16x3_kernel, dim3(cdiv(K, WB_T), cdiv(N, WB_T), S), dim3(256), 0, stream, dy, lddy, x,
                     ldx, M, N, K, rows_per_split, partial, with_bias);
}
