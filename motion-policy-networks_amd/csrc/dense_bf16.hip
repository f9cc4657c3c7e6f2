// dense_bf16.hip -- split-bf16 ("bf16x3") variant of the dense layers: y = act(x . W^T + b) with every fp32
// product evaluated as x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), the same
// arithmetic as sa_mlp_bf16.hip.  3 MFMAs of 32 cycles per 16 k-values instead of 8 fp32 MFMAs of 64 cycles.
// The weights are split once (mpx_split_bf16: two [N, Kp] bf16 planes, Kp = K rounded up to 16, zero padded); the
// activations arrive as fp32 and are split while they are staged into LDS.  128x128 output tile per 256-thread
// workgroup (4 waves as 2x2, 64x64 per wave), K walked 16 at a time through a double-buffered LDS stage, tile order
// XCD-aware like the fp32 kernel.  Opt-in with the other bf16x3 kernels (model.set_precision("bf16x3")).
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

__global__ void __launch_bounds__(256)
    split_bf16_kernel(const float *__restrict__ w, int N, int K, int Kp, __bf16 *__restrict__ hi,
                      __bf16 *__restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * Kp) return;
  const int n = (int)(i / Kp), k = (int)(i - (int64_t)n * Kp);
  const float v = k < K ? w[(size_t)n * K + k] : 0.0f;
  const __bf16 h = (__bf16)v;
  hi[i] = h;
  lo[i] = (__bf16)(v - (float)h);
}

// Store one 32 x 64 block of a wave's results: the accumulators of a 32-row half pass through the wave's own LDS
// staging area (ds_write_b32 in the MFMA layout, back as rows of float4) and leave either as fp32 rows (yh == nullptr)
// or already split into the hi / lo bf16 planes the next layer's plane-input kernel stages without touching them
// (8 bytes per lane and plane, 128 contiguous bytes per row).
constexpr int HB_LDC = 64 + 4;
__device__ __forceinline__ void hb_store_rows(const float *stage, int lane, int row0, int col0, int M, int N, float *y,
                                              int ldy, bool vec_ok, __bf16 *yh, __bf16 *yl, int ldp) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int e = t * 64 + lane;
    const int rr = e >> 4, c4 = (e & 15) * 4;
    const int row = row0 + rr, col = col0 + c4;
    const float4 v = *reinterpret_cast<const float4 *>(&stage[rr * HB_LDC + c4]);
    if (row >= M) continue;
    if (yh != nullptr) {  // (launcher: N and ldp are multiples of 4 here)
      if (col < N) {
        const float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[q] = (__bf16)f[q];
          l[q] = (__bf16)(f[q] - (float)h[q]);
        }
        *reinterpret_cast<bf16x4 *>(yh + (size_t)row * ldp + col) = h;
        *reinterpret_cast<bf16x4 *>(yl + (size_t)row * ldp + col) = l;
      }
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
    linear_bf16x3_kernel(const float *__restrict__ x, int ldx, const __bf16 *__restrict__ wh,
                         const __bf16 *__restrict__ wl, int Kp, const float *__restrict__ bias, int M, int N, int K,
                         int act, float *__restrict__ y, int ldy, __bf16 *__restrict__ yh, __bf16 *__restrict__ yl,
                         int ldp) {
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

  // staging maps: x (fp32) 2 float4 per thread; w planes one 16-byte (8 x bf16) load per thread each
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
      pwh = *reinterpret_cast<const bf16x8 *>(wh + (size_t)n * Kp + k0 + wc);
      pwl = *reinterpret_cast<const bf16x8 *>(wl + (size_t)n * Kp + k0 + wc);
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
  if (POOL) {
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
      hb_store_rows(stage, lane, m0 + wm * 64 + i * 32, n0 + wn * 64, M, N, y, ldy, vec_ok, yh, yl, ldp);
    }
  }
}

// ---- plane-input form: the activations arrive ALREADY split (two bf16 planes [M, lda], written by the layer before)
// so nothing is converted on the way in: all four operand planes of a slab go global -> LDS by DMA
// (`buffer_load_dwordx4 ... lds`, no staging registers, no ds_write, no VALU split -- in the fp32-input kernel above every
// activation is split once per 128-column tile that reads it, N / 128 times).  Slabs of 32 k-values = 64-byte rows; a
// wave's DMA load writes 16 rows x 64 B of consecutive LDS, the 16-byte chunk c of row r sits at physical chunk
// c ^ ((r >> 2) & 3) (the lane picks which global chunk it fetches; the fragment reads apply the same xor: every
// quarter-wave ds_read_b128 touches all 64 banks once -- the layout of dense.hip's DMA kernel).  Same MFMA order per 16
// k-values as the fp32-input kernel, so on planes made by the same split the results are bit-identical to it.
// OUT: 0 = fp32 rows, 1 = hi / lo planes for the next layer, 2 = max over the tile's 128 rows (atomicMax on >= 0 values).
// Ring of STAGES slabs (BK k-values each): the DMA loads of slab kb + STAGES - 1 are issued at the top of slab kb, each wave
// waits (counted vmcnt) only for its own loads of slab kb + 1 before the one barrier of the slab, so the loads of the
// later slabs stay in flight across it.  (`__syncthreads()` would drain them: its fence counts an LDS-DMA as a pending LDS
// write and emits vmcnt(0); the raw s_barrier below is preceded by an explicit lgkmcnt(0) for the fragment reads.)
template <int BK> struct pb_geom {
  static constexpr int ROWB = BK * 2;                 // bytes of a slab row (bf16)
  static constexpr int NCH = BK / 8;                  // 16-byte chunks per row
  static constexpr int RPI = 64 / NCH;                // rows one wave-wide DMA load covers
  static constexpr int SH = BK == 32 ? 2 : 3;         // rows r and r + (256 / ROWB) share banks: xor chunk with r >> SH
  static constexpr int PLANE = HB_BM * ROWB;          // bytes of one operand plane of a stage
  static constexpr int LPS = 4 * (32 / RPI);          // DMA loads per wave and slab
};
template <int OUT, int BK, int STAGES>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(163840 / (STAGES * 4 * 128 * BK * 2) > 4 ? 4 : 163840 / (STAGES * 4 * 128 * BK * 2),
                                       163840 / (STAGES * 4 * 128 * BK * 2) > 4 ? 4 : 163840 / (STAGES * 4 * 128 * BK * 2))))
    linear_bf16x3_planes_kernel(const __bf16 *__restrict__ ah, const __bf16 *__restrict__ al, int lda,
                                const __bf16 *__restrict__ wh, const __bf16 *__restrict__ wl, int Kp,
                                const float *__restrict__ bias, int M, int N, int act, float *__restrict__ y, int ldy,
                                __bf16 *__restrict__ yh, __bf16 *__restrict__ yl, int ldp) {
  typedef pb_geom<BK> G;
  static_assert(STAGES * 4 * G::PLANE <= 65536, "operand ring must fit 64 KB");
  // [stage][plane: A_hi, A_lo, B_hi, B_lo][128 rows x ROWB]
  constexpr int RING_BYTES = STAGES * 4 * G::PLANE, STAGING_BYTES = OUT == 2 ? 0 : 4 * 32 * HB_LDC * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING_BYTES > STAGING_BYTES ? RING_BYTES : STAGING_BYTES];
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

  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t abytes = (uint32_t)((int64_t)M * lda * 2), wbytes = (uint32_t)((int64_t)N * Kp * 2);
  const __amdgpu_buffer_rsrc_t r_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(ah), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(al), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(wh), 0, (int)wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(wl), 0, (int)wbytes, 0x00020000);
  // wave `wave` brings rows [32*wave, 32*wave + 32) of each plane, RPI rows (1 KB) per load: lane -> (row = lane / NCH of
  // the RPI, physical chunk = lane % NCH), fetching logical chunk (lane % NCH) ^ (lane / 16)
  constexpr int NI = 32 / G::RPI;
  int avoff[2], wvoff[2];  // (NI <= 2 used; a dependent array size captured by the lambda below loses the host stub on ROCm 7.2)
  {
    const int c16 = (((lane & (G::NCH - 1)) ^ (lane >> 4)) & (G::NCH - 1)) * 16;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = 32 * wave + G::RPI * i + lane / G::NCH;
      // (a row past the end gets an offset past num_records: the load returns zeros)
      avoff[i] = m0 + r < M ? (int)((uint32_t)(m0 + r) * (uint32_t)lda * 2u + (uint32_t)c16) : (int)0xFFFFFFF0u;
      wvoff[i] = n0 + r < N ? (int)((uint32_t)(n0 + r) * (uint32_t)Kp * 2u + (uint32_t)c16) : (int)0xFFFFFFF0u;
    }
  }
  auto dma = [&](int k0, int stage) __attribute__((always_inline)) {
    unsigned char *base = smem + stage * 4 * G::PLANE + (32 * wave) * G::ROWB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      unsigned char *d = base + G::RPI * i * G::ROWB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_ah, (lds_void *)(d + 0 * G::PLANE), 16, avoff[i], k0 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_al, (lds_void *)(d + 1 * G::PLANE), 16, avoff[i], k0 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wh, (lds_void *)(d + 2 * G::PLANE), 16, wvoff[i], k0 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wl, (lds_void *)(d + 3 * G::PLANE), 16, wvoff[i], k0 * 2, 0, 0);
    }
  };
  // fragment: row r of a plane, logical 16-byte chunk c (k = 8 c .. 8 c + 7 of the slab)
  auto frag = [&](int stage, int p, int r, int c) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8 *>(smem + (stage * 4 + p) * G::PLANE + r * G::ROWB +
                                             (((c ^ (r >> G::SH)) & (G::NCH - 1)) << 4));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  const int nk = Kp / BK;
  constexpr int D = STAGES - 1;                       // slabs in flight
  constexpr int WAIT_NEXT = 0x0070 | ((D - 1) * G::LPS);  // vmcnt((D-1) * LPS), lgkmcnt(0): slab kb + 1 has landed
  static_assert((D - 1) * G::LPS < 16, "vmcnt immediate");
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) dma(d * BK, d);
  if (nk >= D) __builtin_amdgcn_s_waitcnt(WAIT_NEXT); else __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();
  int st = 0, st_in = D % STAGES;
  for (int kb = 0; kb < nk; ++kb) {
    const bool more = kb + D < nk;
    if (more) dma((kb + D) * BK, st_in);
#pragma unroll
    for (int s16 = 0; s16 < BK / 16; ++s16) {  // 16-k steps of the slab; operand k = 8*half + e of the step
      bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + l31, rb = wn * 64 + t * 32 + l31, c = 2 * s16 + half;
        fah[t] = frag(st, 0, ra, c);
        fal[t] = frag(st, 1, ra, c);
        fbh[t] = frag(st, 2, rb, c);
        fbl[t] = frag(st, 3, rb, c);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
        }
    }
    // the fragment reads of this stage are complete (lgkmcnt(0): the stage is refilled right after the barrier) and this
    // wave's part of the next slab has landed; in the tail nothing newer is in flight, so wait for everything
    if (more) __builtin_amdgcn_s_waitcnt(WAIT_NEXT); else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    st = st + 1 == STAGES ? 0 : st + 1;
    st_in = st_in + 1 == STAGES ? 0 : st_in + 1;
  }

  // epilogue: C[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*half
  if (OUT == 2) {
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
    float *stage = reinterpret_cast<float *>(smem) + wave * (32 * HB_LDC);
    const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * half) * HB_LDC + j * 32 + l31] = hb_act(acc[i][j][r] + bv, act);
      }
      hb_store_rows(stage, lane, m0 + wm * 64 + i * 32, n0 + wn * 64, M, N, y, ldy, vec_ok, OUT == 1 ? yh : nullptr, yl,
                    ldp);
    }
  }
}

MPX_EXPORT int mpx_split_bf16(const float *w, int N, int K, void *w_hi, void *w_lo, mpx_stream_t stream) {
  MPX_REQUIRE(N >= 1 && K >= 1 && w && w_hi && w_lo, "mpx_split_bf16: bad argument");
  const int Kp = (K + 15) / 16 * 16;
  hipLaunchKernelGGL(split_bf16_kernel, dim3(cdiv((int64_t)N * Kp, 256)), dim3(256), 0, mpx_s(stream), w, N, K, Kp,
                     reinterpret_cast<__bf16 *>(w_hi), reinterpret_cast<__bf16 *>(w_lo));
  MPX_LAUNCH_CHECK("mpx_split_bf16");
}

static int hb_check(const char *name, const float *x, int ldx, const void *wh, const void *wl, int M, int N, int K,
                    int ldy) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "%s: bad size", name);
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "%s: K and ldx must be multiples of 4 (got %d, %d)", name, K, ldx);
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)wh | (uintptr_t)wl) & 15) == 0, "%s: operands must be 16-byte aligned", name);
  MPX_REQUIRE(ldx >= K && ldy >= N, "%s: leading dimension too small", name);
  return 0;
}

MPX_EXPORT int mpx_linear_bf16x3(const float *x, int ldx, const void *w_hi, const void *w_lo, const float *bias, int M,
                                 int N, int K, int act, float *y, int ldy, mpx_stream_t stream) {
  if (hb_check("mpx_linear_bf16x3", x, ldx, w_hi, w_lo, M, N, K, ldy)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_bf16x3: unknown activation %d", act);
  if (M == 0) return 0;
  MPX_REQUIRE(cdiv(M, HB_BM) <= 65535, "mpx_linear_bf16x3: M too large");
  hipLaunchKernelGGL((linear_bf16x3_kernel<false>), dim3(cdiv(N, HB_BN), cdiv(M, HB_BM)), dim3(256), 0, mpx_s(stream), x,
                     ldx, reinterpret_cast<const __bf16 *>(w_hi), reinterpret_cast<const __bf16 *>(w_lo),
                     (K + 15) / 16 * 16, bias, M, N, K, act, y, ldy, (__bf16 *)nullptr, (__bf16 *)nullptr, 0);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3");
}

static int hb_check_planes_out(const char *name, const void *y_hi, const void *y_lo, int N, int ldp) {
  MPX_REQUIRE(y_hi && y_lo, "%s: output planes missing", name);
  MPX_REQUIRE(N % 4 == 0 && ldp % 4 == 0 && ldp >= N, "%s: N and ldp must be multiples of 4, ldp >= N (got %d, %d)", name, N,
              ldp);
  MPX_REQUIRE((((uintptr_t)y_hi | (uintptr_t)y_lo) & 7) == 0, "%s: output planes must be 8-byte aligned", name);
  return 0;
}

MPX_EXPORT int mpx_linear_bf16x3_to_planes(const float *x, int ldx, const void *w_hi, const void *w_lo, const float *bias,
                                           int M, int N, int K, int act, void *y_hi, void *y_lo, int ldp,
                                           mpx_stream_t stream) {
  if (hb_check("mpx_linear_bf16x3_to_planes", x, ldx, w_hi, w_lo, M, N, K, N)) return 1;
  if (hb_check_planes_out("mpx_linear_bf16x3_to_planes", y_hi, y_lo, N, ldp)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_bf16x3_to_planes: unknown activation %d", act);
  if (M == 0) return 0;
  MPX_REQUIRE(cdiv(M, HB_BM) <= 65535, "mpx_linear_bf16x3_to_planes: M too large");
  hipLaunchKernelGGL((linear_bf16x3_kernel<false>), dim3(cdiv(N, HB_BN), cdiv(M, HB_BM)), dim3(256), 0, mpx_s(stream), x,
                     ldx, reinterpret_cast<const __bf16 *>(w_hi), reinterpret_cast<const __bf16 *>(w_lo),
                     (K + 15) / 16 * 16, bias, M, N, K, act, (float *)nullptr, 0, reinterpret_cast<__bf16 *>(y_hi),
                     reinterpret_cast<__bf16 *>(y_lo), ldp);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3_to_planes");
}

// 32-k slabs x 2 stages (64 KB, two workgroups per CU).  Measured at the group-all shapes, 8192 envs (M = 1 M rows):
// 512 -> 1024 + pooling 3.32 ms (fp32-row input kernel: 3.93), 512 -> 512 2.18 ms (2.43); 16-k slabs were slower than the
// fp32-row kernel whatever the ring depth (2 / 3 / 4 stages at 4 / 3 / 2 workgroups per CU: 4.2-4.3 and 2.4-2.5 ms) --
// half the matrix work per barrier and 32-byte DMA pieces.
#define PB_LAUNCH(OUT, grid, s, ...) \
  hipLaunchKernelGGL((linear_bf16x3_planes_kernel<OUT, 32, 2>), grid, dim3(256), 0, s, __VA_ARGS__)

static int pb_check(const char *name, const void *a_hi, const void *a_lo, int lda, const void *wh, const void *wl, int M,
                    int N, int K) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "%s: bad size", name);
  MPX_REQUIRE(a_hi && a_lo && wh && wl, "%s: operand planes missing", name);
  MPX_REQUIRE(K % 32 == 0 && lda % 8 == 0 && lda >= K, "%s: K must be a multiple of %d and lda a multiple of 8, lda >= K (got %d, %d)",
              name, 32, K, lda);
  MPX_REQUIRE((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)wh | (uintptr_t)wl) & 15) == 0,
              "%s: operand planes must be 16-byte aligned", name);
  MPX_REQUIRE((int64_t)M * lda * 2 < ((int64_t)1 << 32) - 16 && (int64_t)N * K * 2 < ((int64_t)1 << 32) - 16,
              "%s: an operand plane must stay under 4 GB (split the rows over several calls)", name);
  MPX_REQUIRE(cdiv(M, HB_BM) <= 65535, "%s: M too large", name);
  return 0;
}

MPX_EXPORT int mpx_linear_bf16x3_planes(const void *a_hi, const void *a_lo, int lda, const void *w_hi, const void *w_lo,
                                        const float *bias, int M, int N, int K, int act, float *y, int ldy, void *y_hi,
                                        void *y_lo, int ldp, mpx_stream_t stream) {
  if (pb_check("mpx_linear_bf16x3_planes", a_hi, a_lo, lda, w_hi, w_lo, M, N, K)) return 1;
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_bf16x3_planes: unknown activation %d", act);
  MPX_REQUIRE((y != nullptr) != (y_hi != nullptr), "mpx_linear_bf16x3_planes: pass either y (fp32 rows) or y_hi / y_lo (planes)");
  if (y) {
    MPX_REQUIRE(ldy >= N, "mpx_linear_bf16x3_planes: leading dimension too small");
  } else if (hb_check_planes_out("mpx_linear_bf16x3_planes", y_hi, y_lo, N, ldp)) {
    return 1;
  }
  if (M == 0) return 0;
  const dim3 grid(cdiv(N, HB_BN), cdiv(M, HB_BM));
  const __bf16 *ah = reinterpret_cast<const __bf16 *>(a_hi), *al = reinterpret_cast<const __bf16 *>(a_lo);
  const __bf16 *wh = reinterpret_cast<const __bf16 *>(w_hi), *wl = reinterpret_cast<const __bf16 *>(w_lo);
  if (y)
    PB_LAUNCH(0, grid, mpx_s(stream), ah, al, lda, wh, wl, K, bias, M, N, act, y, ldy, (__bf16 *)nullptr, (__bf16 *)nullptr, 0);
  else
    PB_LAUNCH(1, grid, mpx_s(stream), ah, al, lda, wh, wl, K, bias, M, N, act, (float *)nullptr, 0,
                 reinterpret_cast<__bf16 *>(y_hi), reinterpret_cast<__bf16 *>(y_lo), ldp);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3_planes");
}

MPX_EXPORT int mpx_linear_rowmax_bf16x3_planes(const void *a_hi, const void *a_lo, int lda, const void *w_hi,
                                               const void *w_lo, const float *bias, int M, int N, int K, int rows,
                                               float *y, int ldy, mpx_stream_t stream) {
  if (pb_check("mpx_linear_rowmax_bf16x3_planes", a_hi, a_lo, lda, w_hi, w_lo, M, N, K)) return 1;
  MPX_REQUIRE(rows == HB_BM && M % HB_BM == 0, "mpx_linear_rowmax_bf16x3_planes: pooled groups must be exactly %d rows", HB_BM);
  MPX_REQUIRE(y && ldy >= N, "mpx_linear_rowmax_bf16x3_planes: bad output");
  if (M == 0) return 0;
  hipError_t e = hipMemset2DAsync(y, (size_t)ldy * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)(M / HB_BM),
                                  mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_linear_rowmax_bf16x3_planes: memset failed: %s", hipGetErrorString(e));
  PB_LAUNCH(2, dim3(cdiv(N, HB_BN), M / HB_BM), mpx_s(stream), reinterpret_cast<const __bf16 *>(a_hi),
               reinterpret_cast<const __bf16 *>(a_lo), lda, reinterpret_cast<const __bf16 *>(w_hi),
               reinterpret_cast<const __bf16 *>(w_lo), K, bias, M, N, (int)MPX_ACT_RELU, y, ldy, (__bf16 *)nullptr,
               (__bf16 *)nullptr, 0);
  MPX_LAUNCH_CHECK("mpx_linear_rowmax_bf16x3_planes");
}

MPX_EXPORT int mpx_linear_rowmax_bf16x3(const float *x, int ldx, const void *w_hi, const void *w_lo, const float *bias,
                                        int M, int N, int K, int rows, float *y, int ldy, mpx_stream_t stream) {
  if (hb_check("mpx_linear_rowmax_bf16x3", x, ldx, w_hi, w_lo, M, N, K, ldy)) return 1;
  MPX_REQUIRE(rows == HB_BM && M % HB_BM == 0, "mpx_linear_rowmax_bf16x3: pooled groups must be exactly %d rows", HB_BM);
  if (M == 0) return 0;
  MPX_REQUIRE(M / HB_BM <= 65535, "mpx_linear_rowmax_bf16x3: M too large");
  hipError_t e = hipMemset2DAsync(y, (size_t)ldy * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)(M / HB_BM),
                                  mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_linear_rowmax_bf16x3: memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((linear_bf16x3_kernel<true>), dim3(cdiv(N, HB_BN), M / HB_BM), dim3(256), 0, mpx_s(stream), x, ldx,
                     reinterpret_cast<const __bf16 *>(w_hi), reinterpret_cast<const __bf16 *>(w_lo),
                     (K + 15) / 16 * 16, bias, M, N, K, MPX_ACT_RELU, y, ldy, (__bf16 *)nullptr, (__bf16 *)nullptr, 0);
  MPX_LAUNCH_CHECK("mpx_linear_rowmax_bf16x3");
}
