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

template <bool POOL>
__global__ void __launch_bounds__(256)
    linear_bf16x3_kernel(const float *__restrict__ x, int ldx, const __bf16 *__restrict__ wh,
                         const __bf16 *__restrict__ wl, int Kp, const float *__restrict__ bias, int M, int N, int K,
                         int act, float *__restrict__ y, int ldy) {
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
    constexpr int LDC = 64 + 4;
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
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = t * 64 + lane;
        const int rr = e >> 4, c4 = (e & 15) * 4;
        const int row = m0 + wm * 64 + i * 32 + rr;
        const int col = n0 + wn * 64 + c4;
        const float4 v = *reinterpret_cast<const float4 *>(&stage[rr * LDC + c4]);
        if (row < M) {
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
                     (K + 15) / 16 * 16, bias, M, N, K, act, y, ldy);
  MPX_LAUNCH_CHECK("mpx_linear_bf16x3");
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
                     (K + 15) / 16 * 16, bias, M, N, K, MPX_ACT_RELU, y, ldy);
  MPX_LAUNCH_CHECK("mpx_linear_rowmax_bf16x3");
}
