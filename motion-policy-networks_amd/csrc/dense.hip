// dense.hip -- dense layers of the policy: y = act(x . W^T + b) on the exact-fp32 matrix cores,
// GroupNorm + LeakyReLU, and the row-max pooling of the group-all set-abstraction module.
//
// Reference layers (all torch nn.Linear / Conv2d 1x1 / GroupNorm in the reference):
//   MPiNetsPointNet SA3 mlp [256(+3),512,512,1024] + max over the 128 points   model.py:383
//   fc_layer 1024->4096 GN(16) LeakyReLU ->2048 GN(16) LeakyReLU ->2048        model.py:385-393
//   feature_encoder 7->32->64->128->128->64                                     model.py:47-57
//   decoder 2112->512->256->128->7                                              model.py:58-66
//
// GEMM: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 2x2 v_mfma_f32_32x32x2_f32 tiles), K walked 16 at a time through a double-buffered LDS stage
// (register prefetch of slab k+1 while slab k feeds the matrix pipe, one barrier per slab).
// Within a slab lane-half h consumes k = 8h..8h+7, so each lane fetches its 8 operands of a
// tile row with two 16-byte LDS reads; rows are padded to 20 floats, which makes those reads
// conflict-free for the ds_read_b128 lane groups.  fp32 in, fp32 accumulate (bit-identical to
// an fmaf chain in this k order): bound by the 157 TFLOP/s fp32 MFMA pipe.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == MPX_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == MPX_ACT_LEAKY) return v >= 0.0f ? v : v * 0.01f;
  return v;
}

// POOL: instead of storing the [M,N] result, max-pool it over each workgroup's BM = 128 rows (one
// environment of the group-all module) into y[blockIdx.y][n] with atomicMax on the float bits -- valid
// because the pooled values are post-ReLU (>= 0) and y is zero-initialised by the launcher.
// (DMA: four waves per SIMD, 108 VGPRs with the accumulators in them, four 34 KB workgroups per CU; the
// register-staged form needs 140 registers: three)
//
// DMA (K a multiple of the slab, operands under 4 GB, no split): the next slab goes global -> LDS directly
// (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write; the staging instructions were measured to cost
// 11 % of the matrix pipe).  A wave's DMA load writes 1 KB of consecutive LDS -- 16 rows x 64 B, no padding possible
// -- so the 16-byte k-chunk c of row r is kept at physical chunk c ^ ((r >> 2) & 3): the lane picks WHICH global chunk
// it fetches, the fragment reads apply the same xor, and every quarter-wave ds_read_b128 still touches all 64 banks
// once.  Rows past M / N come back as zeros through the buffer descriptor's range check.
template <int BK, bool POOL, bool DMA>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DMA ? 4 : 3, DMA ? 4 : 3)))
    linear_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ w, int ldw,
                  const float *__restrict__ bias, int M, int N, int K, int act, float *__restrict__ y, int ldy,
                  int kslice, size_t zstride, const float *__restrict__ dact_of, int lddact, int dact,
                  const int32_t *__restrict__ seg, int seg_row0) {
  static_assert(!DMA || BK == 16, "the DMA layout is written for 16-float slabs");
  constexpr int LDT = DMA ? BK : BK + 4;  // padded row (or xor-swizzled chunks): conflict-free 16-byte fragment reads
  constexpr int HK = BK / 2;        // k-values per lane-half per slab
  constexpr int NLD = BK / 8;       // float4 staged per thread per matrix per slab
  constexpr int SMEM_OPERANDS = 2 * (BM + BN) * LDT, SMEM_EPILOGUE = POOL ? 0 : 4 * 32 * (64 + 4);
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_OPERANDS > SMEM_EPILOGUE ? SMEM_OPERANDS : SMEM_EPILOGUE];  // [As0 | As1 | Bs0 | Bs1]
  float(*As)[BM * LDT] = reinterpret_cast<float(*)[BM * LDT]>(smem);
  float(*Bs)[BN * LDT] = reinterpret_cast<float(*)[BN * LDT]>(smem + 2 * BM * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.
  // All N-tiles of one 128-row block are given to ONE XCD, so a block of x is fetched from HBM once instead of
  // once per XCD that happens to hold one of its N-tiles (measured 8x over-fetch on the N = 1024 layer).
  int bx = blockIdx.x, by = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7, slot = lin >> 3;
    by = (int)((slot / gridDim.x) * 8 + xcd);
    bx = (int)(slot % gridDim.x);
  }
  const int m0 = by * BM, n0 = bx * BN;
  // split-K launches (gridDim.z > 1): slice z owns k in [z*kslice, (z+1)*kslice) and writes its raw partial
  // tile to y + z*zstride; the launcher passes bias = nullptr / act = none and reduces the slices afterwards.
  {
    const int kz = blockIdx.z * kslice;
    x += kz;
    w += kz;
    y += blockIdx.z * zstride;
    K = min(kslice, K - kz);
  }

  // staging map: thread -> (row, 4-float column chunk); BK/4 chunks per row, NLD rows per thread
  constexpr int CPR = BK / 4, RPP = 256 / CPR;  // chunks per row, rows per pass
  const int srow = tid / CPR, scol = (tid % CPR) * 4;
  float4 pa[NLD], pb[NLD];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int r = srow + RPP * i;
      const int kk = k0 + scol;
      pa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < M && kk < K) pa[i] = *reinterpret_cast<const float4 *>(x + (size_t)(m0 + r) * ldx + kk);
      if (n0 + r < N && kk < K) pb[i] = *reinterpret_cast<const float4 *>(w + (size_t)(n0 + r) * ldw + kk);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int r = srow + RPP * i;
      *reinterpret_cast<float4 *>(&As[buf][r * LDT + scol]) = pa[i];
      *reinterpret_cast<float4 *>(&Bs[buf][r * LDT + scol]) = pb[i];
    }
  };

  // DMA staging: wave `wave` brings rows [32*wave, 32*wave + 32) of both operand tiles, 16 rows (1 KB) per load;
  // lane -> (row = lane / 4 of the 16, physical chunk = lane % 4), fetching logical chunk (lane % 4) ^ (lane / 16)
  typedef __attribute__((address_space(3))) void lds_void;
  __amdgpu_buffer_rsrc_t xrsrc, wrsrc;
  int xvoff[2], wvoff[2];
  if constexpr (DMA) {
    xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)(uint32_t)((int64_t)M * ldx * 4), 0x00020000);
    wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, (int)(uint32_t)((int64_t)N * ldw * 4), 0x00020000);
    const int c4 = (((lane & 3) ^ (lane >> 4)) & 3) * 16;  // byte offset of the logical chunk inside the slab row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * wave + 16 * i + (lane >> 2);
      // (a row past the end gets an offset past num_records: the load returns zeros)
      xvoff[i] = m0 + r < M ? (int)((uint32_t)(m0 + r) * (uint32_t)ldx * 4u + (uint32_t)c4) : (int)0xFFFFFFF0u;
      wvoff[i] = n0 + r < N ? (int)((uint32_t)(n0 + r) * (uint32_t)ldw * 4u + (uint32_t)c4) : (int)0xFFFFFFF0u;
    }
  }
  auto dma = [&](int k0, int buf) __attribute__((always_inline)) {
    if constexpr (!DMA) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void *)&As[buf][(32 * wave + 16 * i) * LDT], 16, xvoff[i], k0 * 4, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void *)&Bs[buf][(32 * wave + 16 * i) * LDT], 16, wvoff[i], k0 * 4, 0, 0);
    }
  };
  // fragment address: row r, 16-byte chunk c of the slab row
  auto frag = [&](int r, int c) __attribute__((always_inline)) { return r * LDT + (DMA ? ((c ^ (r >> 2)) & 3) * 4 : c * 4); };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  const int nk = (K + BK - 1) / BK;  // the k-guards zero-fill a partial last slab
  if constexpr (DMA) {
    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the compiler does not know these loads write LDS
  } else {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  for (int kb = 0; kb < nk; ++kb) {
    const int buf = kb & 1;
    if constexpr (DMA) {
      if (kb + 1 < nk) dma((kb + 1) * BK, buf ^ 1);
    } else {
      if (kb + 1 < nk) gload((kb + 1) * BK);
    }
    float4 a[2][HK / 4], b[2][HK / 4];
#pragma unroll
    for (int v = 0; v < HK / 4; ++v)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t][v] = *reinterpret_cast<const float4 *>(&As[buf][frag(wm * 64 + t * 32 + l31, (HK / 4) * half + v)]);
        b[t][v] = *reinterpret_cast<const float4 *>(&Bs[buf][frag(wn * 64 + t * 32 + l31, (HK / 4) * half + v)]);
      }
#pragma unroll
    for (int v = 0; v < HK / 4; ++v)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float av = u == 0 ? a[i][v].x : (u == 1 ? a[i][v].y : (u == 2 ? a[i][v].z : a[i][v].w));
            const float bv = u == 0 ? b[j][v].x : (u == 1 ? b[j][v].y : (u == 2 ? b[j][v].z : b[j][v].w));
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    // Issue order of the slab.  Left alone the scheduler clusters the eight fragment reads and the first MFMA waits
    // for the last of them; this pipeline starts the matrix pipe after the first half of the reads and feeds the
    // rest in between MFMAs (measured at the model's shapes, 8192 envs: 20.3 -> 19.4 ms per step of dense layers;
    // 1, 3 or 4 MFMAs per read instead of 2: the same within 1 %; reads pinned after the 4th MFMA: 19.9 ms).
    __builtin_amdgcn_sched_group_barrier(0x100, NLD * 2, 0);  // DS reads
#pragma unroll
    for (int i = 0; i < NLD * 2; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one more DS read
    }
    if constexpr (DMA) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // the next slab has landed in LDS (vmcnt(0))
    } else {
      if (kb + 1 < nk) sstore(buf ^ 1);
    }
    __syncthreads();
  }

  // epilogue: C[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*half
  if (POOL && seg != nullptr) {  // max over each SEGMENT of rows (training; y = the 64-bit keys, ldy in keys: common.h)
    float bvj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      bvj[j] = (bias && col < N) ? bias[col] : 0.0f;
    }
    mpx_segpool_tile([&](int i, int j, int r) __attribute__((always_inline)) { return act_apply(acc[i][j][r] + bvj[j], act); },
                     m0 + wm * 64, n0 + wn * 64, M, N, half, l31, seg, seg_row0, reinterpret_cast<unsigned long long *>(y), ldy);
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + l31;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.0f;
    if (POOL) {
      float m = 0.0f;  // post-ReLU values are >= 0
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < M) m = fmaxf(m, act_apply(acc[i][j][r] + bv, act));
        }
      m = mpx_max_across_halves(m);
      if (half == 0) atomicMax(reinterpret_cast<int *>(y + (size_t)by * ldy + col), __float_as_int(m));
    }
  }
  if (!POOL) {
    // Store epilogue staged through LDS (the operand buffers are free after the last barrier): 64 scalar
    // 4-byte stores per lane are store-issue bound; written as rows of float4 it is 16 wide stores.
    // Per wave and per 32-row half: acc -> lds[32][64+4] (ds_write_b32), back as float4 along the row.
    constexpr int LDC = 64 + 4;
    float *stage = smem + wave * (32 * LDC);  // 4 waves x 8.5 KB inside the 40 KB of operand buffers
    static_assert(POOL || 4 * 32 * LDC <= (int)(sizeof(smem) / sizeof(float)), "staging must fit the shared array");
    const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * half) * LDC + j * 32 + l31] = act_apply(acc[i][j][r] + bv, act);
      }
      // (a wave only touches its own staging area: no barrier needed, LDS ops of a wave are ordered)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = t * 64 + lane;      // 512 float4 = 32 rows x 16
        const int rr = e >> 4, c4 = (e & 15) * 4;
        const int row = m0 + wm * 64 + i * 32 + rr;
        const int col = n0 + wn * 64 + c4;
        float4 v = *reinterpret_cast<const float4 *>(&stage[rr * LDC + c4]);
        if (row < M) {
          if (dact_of != nullptr) {  // backward of the layer below (mpx_linear_dact): times act'(its output), as mpx_act_backward does
            const float *mk = dact_of + (size_t)row * lddact + col;
            const float m0_ = col + 0 < N ? mk[0] : 0.0f, m1_ = col + 1 < N ? mk[1] : 0.0f;
            const float m2_ = col + 2 < N ? mk[2] : 0.0f, m3_ = col + 3 < N ? mk[3] : 0.0f;
            if (dact == MPX_ACT_RELU) {
              v.x = m0_ > 0.0f ? v.x : 0.0f, v.y = m1_ > 0.0f ? v.y : 0.0f, v.z = m2_ > 0.0f ? v.z : 0.0f, v.w = m3_ > 0.0f ? v.w : 0.0f;
            } else if (dact == MPX_ACT_LEAKY) {
              v.x = m0_ >= 0.0f ? v.x : 0.01f * v.x, v.y = m1_ >= 0.0f ? v.y : 0.01f * v.y;
              v.z = m2_ >= 0.0f ? v.z : 0.01f * v.z, v.w = m3_ >= 0.0f ? v.w : 0.01f * v.w;
            }
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
    }
  }
}

// the direct-to-LDS variant wants whole slabs and operands a 32-bit buffer offset can span
static bool dma_ok(int M, int N, int K, int ldx, int ldw) {
  const int64_t lim = ((int64_t)1 << 32) - 4096;  // (offsets and num_records are unsigned 32-bit)
  return K % 16 == 0 && (int64_t)M * ldx * 4 < lim && (int64_t)N * ldw * 4 < lim;
}

// ---- a few rows (M <= 8: the head of a single-problem rollout) ---------------------------------------------
// No tile of the matrix pipe is worth filling: the layer is a stream of N*K weights (67 MB for the three fc layers)
// against 1-8 activation rows.  The rows sit in LDS; a wave owns R = 2 output columns and walks their weight rows
// 1 KB at a time (16 bytes per lane, coalesced), one fmaf chain per (row, column) and lane, then a butterfly sum
// over the lanes -- a fixed order, so the result is deterministic (it differs from the MFMA kernel's by fp32
// summation order only).
constexpr int GEMV_R = 2;
template <int MT>
__global__ void __launch_bounds__(256)
    gemv_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ w, const float *__restrict__ bias,
                int M, int N, int K, int act, float *__restrict__ y, int ldy) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [MT][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid * 4; i < MT * K; i += 1024) {  // K % 4 == 0: a float4 never straddles two rows
    const int m = i / K, k = i - m * K;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) v = *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + k);
    *reinterpret_cast<float4 *>(xs + i) = v;
  }
  __syncthreads();
  const int n0 = (blockIdx.x * 4 + wave) * GEMV_R;
  if (n0 >= N) return;
  float acc[MT][GEMV_R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < GEMV_R; ++r) acc[m][r] = 0.0f;
  const float *wr[GEMV_R];
#pragma unroll
  for (int r = 0; r < GEMV_R; ++r) wr[r] = w + (size_t)min(n0 + r, N - 1) * K;  // (a column past N repeats the last: not stored)
#pragma unroll 4
  for (int k = lane * 4; k < K; k += 256) {
    float4 wv[GEMV_R];
#pragma unroll
    for (int r = 0; r < GEMV_R; ++r) wv[r] = *reinterpret_cast<const float4 *>(wr[r] + k);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 xv = *reinterpret_cast<const float4 *>(xs + m * K + k);
#pragma unroll
      for (int r = 0; r < GEMV_R; ++r) {
        float a = acc[m][r];
        a = mpx_fma(xv.x, wv[r].x, a);
        a = mpx_fma(xv.y, wv[r].y, a);
        a = mpx_fma(xv.z, wv[r].z, a);
        a = mpx_fma(xv.w, wv[r].w, a);
        acc[m][r] = a;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < GEMV_R; ++r) {
      float v = acc[m][r];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && m < M && n0 + r < N)
        y[(size_t)m * ldy + n0 + r] = act_apply(v + (bias ? bias[n0 + r] : 0.0f), act);
    }
}

static bool gemv_fits(int M, int K) { return M <= 8 && (int64_t)(M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8) * K * 4 <= 64 * 1024; }

static void gemv_launch(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K, int act,
                        float *y, int ldy, mpx_stream_t stream) {
  const dim3 g(cdiv(N, 4 * GEMV_R)), t(256);
#define GEMV_GO(MT) \
  hipLaunchKernelGGL(gemv_kernel<MT>, g, t, (size_t)MT * K * sizeof(float), mpx_s(stream), x, ldx, w, bias, M, N, K, act, y, ldy)
  if (M <= 1) GEMV_GO(1);
  else if (M <= 2) GEMV_GO(2);
  else if (M <= 4) GEMV_GO(4);
  else GEMV_GO(8);
#undef GEMV_GO
}

// ---- a row per lane: y[M, 128] = x[M, K] . w[128, K]^T for a SHORT K (the second module's per-point first layer: 4.2 M
// rows of [f | xyz | 0], K = 68) ------------------------------------------------------------------------------------
// The tiled kernel above spends that launch on its prologue / epilogue: 5 slabs of K per 128 x 128 tile, a 64 KB tile
// written per 35 KB read -- 1.15 ms for 3.3 GB of traffic.  Here the layer is evaluated the way the fused grouped-MLP
// kernels evaluate theirs (H^T = W . X^T, sa_mlp.hip): a lane owns ONE row, reads its half of it (K / 2 consecutive floats:
// lanes 0-31 the first half of rows 0-31, lanes 32-63 the second) straight into the registers that are the B operands of the
// k-steps, W sits in registers for the whole kernel (K / 2 x 4 values per lane), nothing goes through LDS and there is no
// barrier.  One wave per SIMD, a static stride of 32-row tiles per wave, the next tile's rows requested before the current
// tile's 2 K matrix instructions.  The k-order of a row's sum is (k, k + K / 2) pairs in ascending k -- a different rounding
// order than the tiled kernel's, the same for every M.
template <int KH>
#ifndef MPX_ROWLANE_WAVES
#define MPX_ROWLANE_WAVES 1  // (two waves per SIMD: 1.11 ms instead of 0.71 -- the register budget halves and the row buffers spill)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MPX_ROWLANE_WAVES, MPX_ROWLANE_WAVES)))
    linear_rowlane_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ w, int64_t M, float *__restrict__ y,
                          int ldy) {
  constexpr int K = 2 * KH, NT = 4;  // 4 output tiles of 32 channels
  static_assert(KH % 2 == 0, "a lane reads its half row as 8-byte pieces");
  const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
  const int64_t tiles = (M + 31) / 32;
  // W: A operand of k-step s, output tile ot = w[32 ot + col][KH half + s]
  float wr[KH][NT];
#pragma unroll
  for (int ot = 0; ot < NT; ++ot) {
    const float2 *src = reinterpret_cast<const float2 *>(w + (size_t)(32 * ot + col) * K + KH * half);
#pragma unroll
    for (int s = 0; s < KH / 2; ++s) {
      const float2 v = src[s];
      wr[2 * s][ot] = v.x;
      wr[2 * s + 1][ot] = v.y;
    }
  }
  float cur[KH], nxt[KH];
  auto load_rows = [&](int64_t t, float (&dst)[KH]) __attribute__((always_inline)) {
    int64_t row = t * 32 + col;
    row = row < M ? row : M - 1;  // (rows past the end: a valid row's data, never stored)
    const float2 *src = reinterpret_cast<const float2 *>(x + row * ldx + KH * half);
#pragma unroll
    for (int s = 0; s < KH / 2; ++s) {
      const float2 v = src[s];
      dst[2 * s] = v.x;
      dst[2 * s + 1] = v.y;
    }
  };
  int64_t t = blockIdx.x;
  if (t < tiles) load_rows(t, cur);
  for (; t < tiles; t += gridDim.x) {
    const int64_t tn = t + gridDim.x;
    if (tn < tiles) load_rows(tn, nxt);
    f32x16 acc[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) acc[ot] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
      for (int ot = 0; ot < NT; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[s][ot], cur[s], acc[ot], 0, 0, 0);
    // register r of tile ot = channel 32 ot + 8 (r >> 2) + 4 half + (r & 3) of row `col`: four 16-byte pieces per tile
    const int64_t row = t * 32 + col;
    if (row < M) {
      float *dst = y + row * ldy + 4 * half;
#pragma unroll
      for (int ot = 0; ot < NT; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(dst + 32 * ot + 8 * g) =
              make_float4(acc[ot][4 * g], acc[ot][4 * g + 1], acc[ot][4 * g + 2], acc[ot][4 * g + 3]);
    }
    if (tn < tiles) {
#pragma unroll
      for (int s = 0; s < KH; ++s) cur[s] = nxt[s];
    }
  }
}
static bool rowlane_ok(const float *x, int ldx, const float *w, const float *bias, int N, int K, int act, const float *y, int ldy) {
  return N == 128 && K == 68 && bias == nullptr && act == MPX_ACT_NONE && ldx % 2 == 0 && ldy % 4 == 0 &&
         (((uintptr_t)x | (uintptr_t)w) & 7) == 0 && ((uintptr_t)y & 15) == 0;
}

MPX_EXPORT int mpx_linear(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K,
                          int act, float *y, int ldy, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "mpx_linear: bad size");
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "mpx_linear: K and ldx must be multiples of 4 (got %d, %d)", K, ldx);
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "mpx_linear: x and w must be 16-byte aligned");
  MPX_REQUIRE(ldx >= K && ldy >= N, "mpx_linear: leading dimension too small");
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear: unknown activation %d", act);
  if (M == 0) return 0;
  if (gemv_fits(M, K)) {
    gemv_launch(x, ldx, w, bias, M, N, K, act, y, ldy, stream);
    MPX_LAUNCH_CHECK("mpx_linear");
  }
  if (rowlane_ok(x, ldx, w, bias, N, K, act, y, ldy)) {  // short K, 128 outputs: a row per lane, W in registers (every M: one rounding order)
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const int64_t tiles = ((int64_t)M + 31) / 32, waves = (int64_t)cus * 4 * MPX_ROWLANE_WAVES;
    hipLaunchKernelGGL((linear_rowlane_kernel<34>), dim3((unsigned)(tiles < waves ? tiles : waves)), dim3(64), 0, mpx_s(stream), x,
                       ldx, w, (int64_t)M, y, ldy);
    MPX_LAUNCH_CHECK("mpx_linear");
  }
  if (const int64_t slab = mpx_row_slab(BM, (int64_t)ldx * 4); M > slab) {  // more rows than one launch covers
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear(x + m0 * ldx, ldx, w, bias, (int)(M - m0 < slab ? M - m0 : slab), N, K, act, y + m0 * ldy, ldy, stream))
        return rc;
    return 0;
  }
  // (BK = 32 slabs were measured: no gain, twice the LDS)
  // (A 256 x 128-tile form with a three-stage counted-vmcnt ring at two waves per SIMD -- the structure of
  // dense_bf16.hip's pairs kernel -- was measured on the 1 M-row layers: 8.98 vs 8.32 ms and 4.56 vs 4.21 ms: with
  // 64-cycle fp32 MFMAs four waves per SIMD cover more than the leaner slab does.)
  if (dma_ok(M, N, K, ldx, K))
    hipLaunchKernelGGL((linear_kernel<16, false, true>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 0, mpx_s(stream), x,
                       ldx, w, K, bias, M, N, K, act, y, ldy, K, (size_t)0, static_cast<const float *>(nullptr), 0, 0, static_cast<const int32_t *>(nullptr), 0);
  else
    hipLaunchKernelGGL((linear_kernel<16, false, false>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 0, mpx_s(stream), x,
                       ldx, w, K, bias, M, N, K, act, y, ldy, K, (size_t)0, static_cast<const float *>(nullptr), 0, 0, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear");
}

// dX of the layer above, already multiplied by act'(output) of the layer below: y = (x . w^T) * act'(dact_of) -- the
// input-gradient GEMM of a dense layer (x = dZ of the layer, w = its weights transposed) with the elementwise backward
// of the previous layer's activation in the epilogue (row N1; mpx_act_backward's arithmetic, one pass over the rows
// instead of a GEMM store + a read-modify-write).  Every M takes the 128 x 128 tile kernel.
MPX_EXPORT int mpx_linear_dact(const float *x, int ldx, const float *w, int M, int N, int K, const float *dact_of,
                               int lddact, int dact, float *y, int ldy, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "mpx_linear_dact: bad size");
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "mpx_linear_dact: K and ldx must be multiples of 4 (got %d, %d)", K, ldx);
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "mpx_linear_dact: x and w must be 16-byte aligned");
  MPX_REQUIRE(ldx >= K && ldy >= N, "mpx_linear_dact: leading dimension too small");
  MPX_REQUIRE(dact == MPX_ACT_NONE || (dact_of != nullptr && lddact >= N && (dact == MPX_ACT_RELU || dact == MPX_ACT_LEAKY)),
              "mpx_linear_dact: the activation's output rows are missing or too short, or the activation is unknown");
  if (M == 0) return 0;
  if (dact == MPX_ACT_NONE) dact_of = nullptr;
  if (const int64_t slab = mpx_row_slab(BM, (int64_t)ldx * 4); M > slab) {  // more rows than one launch covers
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_dact(x + m0 * ldx, ldx, w, (int)(M - m0 < slab ? M - m0 : slab), N, K,
                                   dact_of ? dact_of + m0 * lddact : nullptr, lddact, dact, y + m0 * ldy, ldy, stream))
        return rc;
    return 0;
  }
  if (dma_ok(M, N, K, ldx, K))
    hipLaunchKernelGGL((linear_kernel<16, false, true>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 0, mpx_s(stream), x,
                       ldx, w, K, static_cast<const float *>(nullptr), M, N, K, MPX_ACT_NONE, y, ldy, K, (size_t)0, dact_of,
                       lddact, dact, static_cast<const int32_t *>(nullptr), 0);
  else
    hipLaunchKernelGGL((linear_kernel<16, false, false>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 0, mpx_s(stream), x,
                       ldx, w, K, static_cast<const float *>(nullptr), M, N, K, MPX_ACT_NONE, y, ldy, K, (size_t)0, dact_of,
                       lddact, dact, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear_dact");
}

// ---- split-K for skinny problems ------------------------------------------------------------------------
// A rollout of ONE problem (or a few hundred) leaves the fc / decoder layers with 1..32 output tiles: 4096->2048
// at M <= 128 is 16 workgroups walking K = 4096 alone (310 us on 6 % of the CUs).  With a workspace the K range
// is cut into S slices (blockIdx.z), each writing a raw partial tile; splitk_reduce_kernel then adds the S
// partials IN SLICE ORDER (deterministic), the bias and the activation.
static int splitk_plan(int M, int N, int K, int *kslice) {
  const int64_t tiles = (int64_t)cdiv(M, BM) * cdiv(N, BN);
  *kslice = K;
  // (M > 1024: the partial tiles would cost more HBM traffic than the idle CUs are worth)
  if (tiles >= 128 || K < 256 || M > 1024 || gemv_fits(M, K)) return 1;
  const int slabs = cdiv(K, 16);
  int S = (int)((512 + tiles - 1) / tiles);
  if (S > slabs / 4) S = slabs / 4;  // >= 64 k per slice
  if (S < 2) return 1;
  const int per = cdiv(slabs, S);
  *kslice = per * 16;
  return cdiv(K, *kslice);
}

__global__ void __launch_bounds__(256)
    splitk_reduce_kernel(const float *__restrict__ part, int S, size_t zstride, const float *__restrict__ bias,
                         int M, int N, int act, float *__restrict__ y, int ldy) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * N) return;
  const int row = (int)(e / N), col = (int)(e % N);
  float acc = part[e];
  for (int s = 1; s < S; ++s) acc += part[s * zstride + e];
  if (bias) acc += bias[col];
  y[(size_t)row * ldy + col] = act_apply(acc, act);
}

MPX_EXPORT int64_t mpx_linear_workspace(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int kslice;
  const int S = splitk_plan(M, N, K, &kslice);
  return S > 1 ? (int64_t)S * M * N * (int64_t)sizeof(float) : 0;
}

MPX_EXPORT int mpx_linear_ws(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K,
                             int act, float *y, int ldy, void *workspace, int64_t workspace_bytes,
                             mpx_stream_t stream) {
  int kslice = K;
  const int S = (M > 0 && N > 0 && K > 0) ? splitk_plan(M, N, K, &kslice) : 1;
  if (S <= 1 || workspace == nullptr) return mpx_linear(x, ldx, w, bias, M, N, K, act, y, ldy, stream);
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "mpx_linear_ws: K and ldx must be multiples of 4 (got %d, %d)", K, ldx);
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)workspace) & 15) == 0,
              "mpx_linear_ws: x, w and workspace must be 16-byte aligned");
  MPX_REQUIRE(ldx >= K && ldy >= N, "mpx_linear_ws: leading dimension too small");
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_linear_ws: unknown activation %d", act);
  MPX_REQUIRE(workspace_bytes >= mpx_linear_workspace(M, N, K),
              "mpx_linear_ws: workspace of %lld bytes, mpx_linear_workspace asks for %lld", (long long)workspace_bytes,
              (long long)mpx_linear_workspace(M, N, K));
  float *part = static_cast<float *>(workspace);
  const size_t zstride = (size_t)M * N;
  if (dma_ok(M, N, K, ldx, K))  // (slices are multiples of the slab, so every slice keeps whole slabs too)
    hipLaunchKernelGGL((linear_kernel<16, false, true>), dim3(cdiv(N, BN), cdiv(M, BM), S), dim3(256), 0, mpx_s(stream), x,
                       ldx, w, K, static_cast<const float *>(nullptr), M, N, K, MPX_ACT_NONE, part, N, kslice, zstride, static_cast<const float *>(nullptr), 0, 0, static_cast<const int32_t *>(nullptr), 0);
  else
    hipLaunchKernelGGL((linear_kernel<16, false, false>), dim3(cdiv(N, BN), cdiv(M, BM), S), dim3(256), 0, mpx_s(stream), x,
                       ldx, w, K, static_cast<const float *>(nullptr), M, N, K, MPX_ACT_NONE, part, N, kslice, zstride, static_cast<const float *>(nullptr), 0, 0, static_cast<const int32_t *>(nullptr), 0);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((int64_t)M * N, 256)), dim3(256), 0, mpx_s(stream), part, S,
                     zstride, bias, M, N, act, y, ldy);
  MPX_LAUNCH_CHECK("mpx_linear_ws");
}

MPX_EXPORT int mpx_linear_rowmax(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K,
                                 int rows, float *y, int ldy, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "mpx_linear_rowmax: bad size");
  MPX_REQUIRE(rows == BM && M % BM == 0, "mpx_linear_rowmax: pooled groups must be exactly %d rows", BM);
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "mpx_linear_rowmax: K and ldx must be multiples of 4");
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "mpx_linear_rowmax: x and w must be 16-byte aligned");
  MPX_REQUIRE(ldx >= K && ldy >= N, "mpx_linear_rowmax: leading dimension too small");
  if (M == 0) return 0;
  if (const int64_t slab = mpx_row_slab(BM, (int64_t)ldx * 4); M > slab) {
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = mpx_linear_rowmax(x + m0 * ldx, ldx, w, bias, (int)(M - m0 < slab ? M - m0 : slab), N, K, rows,
                                     y + (m0 / BM) * ldy, ldy, stream))
        return rc;
    return 0;
  }
  hipError_t e = hipMemset2DAsync(y, (size_t)ldy * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)(M / BM),
                                  mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_linear_rowmax: memset failed: %s", hipGetErrorString(e));
  if (dma_ok(M, N, K, ldx, K))
    hipLaunchKernelGGL((linear_kernel<16, true, true>), dim3(cdiv(N, BN), M / BM), dim3(256), 0, mpx_s(stream), x, ldx, w,
                       K, bias, M, N, K, MPX_ACT_RELU, y, ldy, K, (size_t)0, static_cast<const float *>(nullptr), 0, 0, static_cast<const int32_t *>(nullptr), 0);
  else
    hipLaunchKernelGGL((linear_kernel<16, true, false>), dim3(cdiv(N, BN), M / BM), dim3(256), 0, mpx_s(stream), x, ldx, w,
                       K, bias, M, N, K, MPX_ACT_RELU, y, ldy, K, (size_t)0, static_cast<const float *>(nullptr), 0, 0, static_cast<const int32_t *>(nullptr), 0);
  MPX_LAUNCH_CHECK("mpx_linear_rowmax");
}

// ---- last layer of a grouped MLP + activation + max over each query's rows (training, row N1) -------------------------
// pooled[q, n] = max over the rows r of segment q of act(x[r] . w[n] + b[n]), arg[q, n] = the first such row -- what
// mpx_linear + mpx_segment_max give, bit for bit, without the [M, N] matrix in memory (2 GB for the second module at
// batch 256): the tile kernel's epilogue folds its 128 x 128 tile into 64-bit {value, ~row} keys (common.h), unpacked here.
__global__ void __launch_bounds__(256)
    segmax_unpack_kernel(const unsigned long long *__restrict__ keys, int64_t Q, int N, float *__restrict__ pooled, int ldp,
                         int64_t *__restrict__ arg) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= Q * N) return;
  const int64_t q = e / N;
  const int n = (int)(e - q * N);
  const unsigned long long k = keys[e];
  pooled[q * ldp + n] = mpx_ordered_float((unsigned)(k >> 32));
  arg[e] = (int64_t)(0xFFFFFFFFu - (unsigned)k);
}
int mpx_segmax_unpack_launch(const unsigned long long *keys, int64_t Q, int N, float *pooled, int ldp, int64_t *arg,
                             hipStream_t stream) {  // (dense_bf16.hip shares it)
  hipLaunchKernelGGL(segmax_unpack_kernel, dim3(cdiv(Q * N, 256)), dim3(256), 0, stream, keys, Q, N, pooled, ldp, arg);
  return 0;
}
int mpx_segmax_check(const char *name, int M, const int32_t *seg, int64_t Q, int N, const void *keys, const float *pooled,
                     int ldp, const int64_t *arg) {
  MPX_REQUIRE(seg && keys && pooled && arg, "%s: NULL operand", name);
  MPX_REQUIRE(Q >= 1 && Q * N < ((int64_t)1 << 37) && ldp >= N, "%s: bad pooled shape", name);
  MPX_REQUIRE(((uintptr_t)keys & 7) == 0, "%s: keys must be 8-byte aligned", name);
  MPX_REQUIRE(M >= 1, "%s: every segment holds at least one row", name);
  return 0;
}
static int segmax_rows(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K, int act,
                       const int32_t *seg, int row0, unsigned long long *keys, mpx_stream_t stream) {
  if (const int64_t slab = mpx_row_slab(BM, (int64_t)ldx * 4); M > slab) {  // more rows than one launch covers
    for (int64_t m0 = 0; m0 < M; m0 += slab)
      if (int rc = segmax_rows(x + m0 * ldx, ldx, w, bias, (int)(M - m0 < slab ? M - m0 : slab), N, K, act, seg + m0,
                               row0 + (int)m0, keys, stream))
        return rc;
    return 0;
  }
  float *ky = reinterpret_cast<float *>(keys);
  if (dma_ok(M, N, K, ldx, K))
    hipLaunchKernelGGL((linear_kernel<16, true, true>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 0, mpx_s(stream), x, ldx, w,
                       K, bias, M, N, K, act, ky, N, K, (size_t)0, static_cast<const float *>(nullptr), 0, 0, seg, row0);
  else
    hipLaunchKernelGGL((linear_kernel<16, true, false>), dim3(cdiv(N, BN), cdiv(M, BM)), dim3(256), 0, mpx_s(stream), x, ldx, w,
                       K, bias, M, N, K, act, ky, N, K, (size_t)0, static_cast<const float *>(nullptr), 0, 0, seg, row0);
  return 0;
}
MPX_EXPORT int mpx_linear_segmax(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K, int act,
                                 const int32_t *seg, int64_t Q, void *keys, float *pooled, int ldp, int64_t *arg,
                                 mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && N >= 1 && K >= 1, "mpx_linear_segmax: bad size");
  MPX_REQUIRE(K % 4 == 0 && ldx % 4 == 0, "mpx_linear_segmax: K and ldx must be multiples of 4 (got %d, %d)", K, ldx);
  MPX_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "mpx_linear_segmax: x and w must be 16-byte aligned");
  MPX_REQUIRE(ldx >= K && act >= 0 && act <= 2, "mpx_linear_segmax: leading dimension too small / unknown activation");
  if (mpx_segmax_check("mpx_linear_segmax", M, seg, Q, N, keys, pooled, ldp, arg)) return 1;
  hipError_t e = hipMemsetAsync(keys, 0, (size_t)Q * N * 8, mpx_s(stream));
  MPX_REQUIRE(e == hipSuccess, "mpx_linear_segmax: memset failed: %s", hipGetErrorString(e));
  if (int rc = segmax_rows(x, ldx, w, bias, M, N, K, act, seg, 0, static_cast<unsigned long long *>(keys), stream)) return rc;
  mpx_segmax_unpack_launch(static_cast<const unsigned long long *>(keys), Q, N, pooled, ldp, arg, mpx_s(stream));
  MPX_LAUNCH_CHECK("mpx_linear_segmax");
}

// ---- GroupNorm + LeakyReLU ---------------------------------------------------------------------------
// one wave per (row, group); values held in registers between the mean and variance passes.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(256)
    groupnorm_leaky_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                           const float *__restrict__ beta, int64_t n_rg, int C, int groups, float eps,
                           float *__restrict__ y, __bf16 *__restrict__ pairs, int ldp) {
  const int lane = threadIdx.x & 63;
  const int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rg >= n_rg) return;
  const int gs = C / groups;
  const int64_t row = rg / groups;
  const int g = (int)(rg % groups);
  const float *xp = x + row * C + (int64_t)g * gs;
  float *yp = y + row * C + (int64_t)g * gs;
  constexpr int MAXV = 8;  // gs <= 512
  float v[MAXV];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < gs ? xp[c] : 0.0f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)gs;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    const float d = c < gs ? v[i] - mean : 0.0f;
    q += d * d;
  }
  const float var = wave_sum(q) / (float)gs;  // biased, like torch
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < gs) {
      float o = (v[i] - mean) * rstd * gamma[g * gs + c] + beta[g * gs + c];
      o = o >= 0.0f ? o : o * 0.01f;
      if (pairs) {  // the bf16x3 dense kernels' operand form: per 16 channels [hi x 16 | lo x 16] (dense_bf16.hip)
        const int cc = g * gs + c;
        __bf16 *d = pairs + row * (int64_t)ldp + (cc >> 4) * 32 + (cc & 15);
        const __bf16 h = (__bf16)o;
        d[0] = h;
        d[16] = (__bf16)(o - (float)h);
      } else {
        yp[c] = o;
      }
    }
  }
}

MPX_EXPORT int mpx_groupnorm_leaky(const float *x, const float *gamma, const float *beta, int M, int C,
                                   int groups, float eps, float *y, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && C >= 1 && groups >= 1 && C % groups == 0, "mpx_groupnorm_leaky: bad size");
  MPX_REQUIRE(C / groups <= 512, "mpx_groupnorm_leaky: group size %d > 512 unsupported", C / groups);
  if (M == 0) return 0;
  const int64_t n = (int64_t)M * groups;
  hipLaunchKernelGGL(groupnorm_leaky_kernel, dim3(cdiv(n, 4)), dim3(256), 0, mpx_s(stream), x, gamma, beta, n, C,
                     groups, eps, y, (__bf16 *)nullptr, 0);
  MPX_LAUNCH_CHECK("mpx_groupnorm_leaky");
}

// the same, with the result written in the bf16x3 dense kernels' pairs form (mpx_split_bf16) instead of fp32 rows
MPX_EXPORT int mpx_groupnorm_leaky_to_pairs(const float *x, const float *gamma, const float *beta, int M, int C, int groups,
                                            float eps, void *y_pairs, int ldp, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && C >= 1 && groups >= 1 && C % groups == 0, "mpx_groupnorm_leaky_to_pairs: bad size");
  MPX_REQUIRE(C / groups <= 512, "mpx_groupnorm_leaky_to_pairs: group size %d > 512 unsupported", C / groups);
  MPX_REQUIRE(y_pairs && C % 16 == 0 && ldp >= 2 * C, "mpx_groupnorm_leaky_to_pairs: C must be a multiple of 16, ldp >= 2 C");
  if (M == 0) return 0;
  const int64_t n = (int64_t)M * groups;
  hipLaunchKernelGGL(groupnorm_leaky_kernel, dim3(cdiv(n, 4)), dim3(256), 0, mpx_s(stream), x, gamma, beta, n, C,
                     groups, eps, (float *)nullptr, reinterpret_cast<__bf16 *>(y_pairs), ldp);
  MPX_LAUNCH_CHECK("mpx_groupnorm_leaky_to_pairs");
}

// ---- max over groups of consecutive rows -------------------------------------------------------------
// workgroup = 64 columns x 4 row groups (each thread walks every 4th row, coalesced 256-byte reads per row),
// then the four partial maxima meet in LDS
__global__ void __launch_bounds__(256)
    rowmax_kernel(const float *__restrict__ x, int ldx, int rows, int C, float *__restrict__ y, int ldy) {
  __shared__ float part[4][64];
  const int g = blockIdx.y, lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float m = -__builtin_inff();
  if (c < C) {
    const float *p = x + (size_t)g * rows * ldx + c;
    for (int r = rg; r < rows; r += 4) m = fmaxf(m, p[(size_t)r * ldx]);
  }
  part[rg][lane] = m;
  __syncthreads();
  if (rg == 0 && c < C)
    y[(size_t)g * ldy + c] = fmaxf(fmaxf(part[0][lane], part[1][lane]), fmaxf(part[2][lane], part[3][lane]));
}

MPX_EXPORT int mpx_rowmax(const float *x, int ldx, int G, int rows, int C, float *y, int ldy,
                          mpx_stream_t stream) {
  MPX_REQUIRE(G >= 0 && rows >= 1 && C >= 1 && ldx >= C && ldy >= C, "mpx_rowmax: bad size");
  if (G == 0) return 0;
  if (G > MPX_GRID_Y) {
    for (int64_t g0 = 0; g0 < G; g0 += MPX_GRID_Y)
      if (int rc = mpx_rowmax(x + g0 * rows * ldx, ldx, (int)(G - g0 < MPX_GRID_Y ? G - g0 : MPX_GRID_Y), rows, C, y + g0 * ldy, ldy,
                              stream))
        return rc;
    return 0;
  }
  hipLaunchKernelGGL(rowmax_kernel, dim3(cdiv(C, 64), G), dim3(256), 0, mpx_s(stream), x, ldx, rows, C, y, ldy);
  MPX_LAUNCH_CHECK("mpx_rowmax");
}
