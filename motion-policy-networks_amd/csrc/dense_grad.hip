// dense_grad.hip -- backward of the dense layers (row N1 of SURVEY.md section 8f: the training path's GEMMs).
//   dX = dY . W            -> mpx_linear(dY, W^T) (the forward kernel; the caller keeps W^T)
//   dW = dY^T . X          -> mpx_linear_wgrad: a 128 x 128 tile of dW per workgroup, the reduction runs over the ROWS
//                             of the batch (millions for the grouped MLPs), split across gridDim.z; every split
//                             writes its partial tile and mpx_reduce_partials adds them in a fixed order
//                             (deterministic, no atomics).  Same fp32 MFMA inner loop as dense.hip: the 16-row slabs
//                             of dY and X are transposed while they are staged into LDS so both operands are k-major.
//   db = column sums of dY -> accumulated by the first k-tile's workgroups from the slabs they stage anyway
//   dZ = dY * act'(y)      -> mpx_act_backward (ReLU / LeakyReLU masks from the layer's OUTPUT)
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_T = 128, WG_BK = 16, WG_LDT = WG_BK + 4;

__global__ void __launch_bounds__(256)
    linear_wgrad_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ldx, int M, int N,
                        int K, int rows_per_split, float *__restrict__ partial, int with_bias) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * WG_T * WG_LDT];  // [As0 | As1 | Bs0 | Bs1]
  float(*As)[WG_T * WG_LDT] = reinterpret_cast<float(*)[WG_T * WG_LDT]>(smem);
  float(*Bs)[WG_T * WG_LDT] = reinterpret_cast<float(*)[WG_T * WG_LDT]>(smem + 2 * WG_T * WG_LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int k0 = blockIdx.x * WG_T, n0 = blockIdx.y * WG_T;  // tile of dW [N, K]: rows n, columns k
  const int mb = blockIdx.z * rows_per_split, me = min(M, mb + rows_per_split);

  // staging: a slab is 16 batch rows x 128 columns of dY (and of X); thread -> (row r, 4-column chunk c4)
  const int sr = tid >> 5, sc = (tid & 31) * 4;
  float4 pa[2], pb[2];
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + sr + 8 * i;
      pa[i] = pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < me) {
        if (n0 + sc < N) pa[i] = *reinterpret_cast<const float4 *>(dy + (size_t)m * lddy + n0 + sc);
        if (k0 + sc < K) pb[i] = *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + k0 + sc);
      }
    }
  };
  auto sstore = [&](int buf) {  // transposed: LDS row = output index (n or k), LDS column = batch row of the slab
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = sr + 8 * i;
      As[buf][(sc + 0) * WG_LDT + r] = pa[i].x, As[buf][(sc + 1) * WG_LDT + r] = pa[i].y;
      As[buf][(sc + 2) * WG_LDT + r] = pa[i].z, As[buf][(sc + 3) * WG_LDT + r] = pa[i].w;
      Bs[buf][(sc + 0) * WG_LDT + r] = pb[i].x, Bs[buf][(sc + 1) * WG_LDT + r] = pb[i].y;
      Bs[buf][(sc + 2) * WG_LDT + r] = pb[i].z, Bs[buf][(sc + 3) * WG_LDT + r] = pb[i].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // bias gradient = column sums of dY: the workgroups of the first k-tile add up the slabs they stage anyway
  const bool do_bias = with_bias && blockIdx.x == 0 && tid < WG_T;
  float bsum = 0.0f;
  const int nslab = (me - mb + WG_BK - 1) / WG_BK;
  if (nslab > 0) {
    gload(mb);
    sstore(0);
  }
  __syncthreads();
  for (int kb = 0; kb < nslab; ++kb) {
    const int buf = kb & 1;
    if (kb + 1 < nslab) gload(mb + (kb + 1) * WG_BK);
    float4 a[2][2], b[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        a[t][v] = *reinterpret_cast<const float4 *>(&As[buf][(wm * 64 + t * 32 + l31) * WG_LDT + 8 * half + 4 * v]);
        b[t][v] = *reinterpret_cast<const float4 *>(&Bs[buf][(wn * 64 + t * 32 + l31) * WG_LDT + 8 * half + 4 * v]);
      }
#pragma unroll
    for (int v = 0; v < 2; ++v)
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
    if (do_bias) {
#pragma unroll
      for (int v = 0; v < WG_BK / 4; ++v) {
        const float4 q = *reinterpret_cast<const float4 *>(&As[buf][tid * WG_LDT + 4 * v]);
        bsum += (q.x + q.y) + (q.z + q.w);
      }
    }
    if (kb + 1 < nslab) sstore(buf ^ 1);
    __syncthreads();
  }
  // per split: [N*K weight partials | N bias partials]; C[row][col]: col = lane&31 (k), row = (r&3) + 8*(r>>2) + 4*half (n)
  float *dst = partial + (size_t)blockIdx.z * ((size_t)N * K + N);
  if (do_bias && n0 + tid < N) dst[(size_t)N * K + n0 + tid] = bsum;
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

// out[i] = sum_s partial[s * stride + i], s ascending
__global__ void __launch_bounds__(256)
    reduce_partials_kernel(const float *__restrict__ partial, int S, int64_t stride, int64_t n, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.0f;
  for (int s = 0; s < S; ++s) acc += partial[(size_t)s * stride + i];
  out[i] = acc;
}

// dz = dy * act'(y): ReLU -> y > 0; LeakyReLU(0.01) -> y >= 0 ? 1 : 0.01 (sign of the output = sign of the input)
__global__ void __launch_bounds__(256)
    act_backward_kernel(const float *__restrict__ dy, const float *__restrict__ y, int64_t n, int act,
                        float *__restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float g = dy[i], o = y[i];
  dz[i] = act == MPX_ACT_RELU ? (o > 0.0f ? g : 0.0f) : (act == MPX_ACT_LEAKY ? (o >= 0.0f ? g : 0.01f * g) : g);
}

MPX_EXPORT int64_t mpx_linear_wgrad_scratch(int M, int N, int K) {
  const int tiles = cdiv(N, WG_T) * cdiv(K, WG_T);
  int S = cdiv(1024, tiles);
  const int maxs = cdiv(M, 4 * WG_BK);
  S = S < 1 ? 1 : (S > maxs ? (maxs < 1 ? 1 : maxs) : S);
  return (int64_t)S * ((int64_t)N * K + N);  // floats: weight partials + bias partials
}

MPX_EXPORT int mpx_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, float *dw,
                                float *db, float *scratch, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 1 && N >= 1 && K >= 1, "mpx_linear_wgrad: bad size");
  MPX_REQUIRE(N % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0,
              "mpx_linear_wgrad: N, K and the leading dimensions must be multiples of 4");
  MPX_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0, "mpx_linear_wgrad: operands must be 16-byte aligned");
  MPX_REQUIRE(dw && scratch, "mpx_linear_wgrad: NULL output / scratch");
  const int64_t per = (int64_t)N * K + N;
  const int S = (int)(mpx_linear_wgrad_scratch(M, N, K) / per);
  const int rps = cdiv(cdiv(M, S), WG_BK) * WG_BK;
  MPX_REQUIRE(cdiv(N, WG_T) <= 65535 && S <= 65535, "mpx_linear_wgrad: grid too large");
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(cdiv(K, WG_T), cdiv(N, WG_T), S), dim3(256), 0, mpx_s(stream), dy, lddy, x,
                     ldx, M, N, K, rps, scratch, db ? 1 : 0);
  // dw [N*K] and db [N] are adjacent in every split's slice: one reduction (db lands right behind dw if the caller
  // laid them out that way, else two launches)
  if (db == dw + (size_t)N * K) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(per, 256)), dim3(256), 0, mpx_s(stream), scratch, S, per, per, dw);
  } else {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv((int64_t)N * K, 256)), dim3(256), 0, mpx_s(stream), scratch, S,
                       per, (int64_t)N * K, dw);
    if (db)
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(N, 256)), dim3(256), 0, mpx_s(stream), scratch + (size_t)N * K,
                         S, per, (int64_t)N, db);
  }
  MPX_LAUNCH_CHECK("mpx_linear_wgrad");
}

MPX_EXPORT int mpx_act_backward(const float *dy, const float *y, int64_t n, int act, float *dz, mpx_stream_t stream) {
  MPX_REQUIRE(n >= 0 && act >= 0 && act <= 2, "mpx_act_backward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_backward_kernel, dim3(cdiv(n, 256)), dim3(256), 0, mpx_s(stream), dy, y, n, act, dz);
  MPX_LAUNCH_CHECK("mpx_act_backward");
}
