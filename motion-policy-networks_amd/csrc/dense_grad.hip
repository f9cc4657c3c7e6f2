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

constexpr int WG_BK = 16, WG_LDT = WG_BK + 4;

// T = 128: a 128 x 128 tile of dW per workgroup (four waves as 2 x 2, 64 x 64 each); T = 64 (layers of <= 64 inputs and
// outputs -- the first set-abstraction module, whose three weight gradients reduce over ~2 M rows): a 64 x 64 tile, one
// 32 x 32 MFMA tile per wave -- on the 128-wide kernel three quarters of such a layer's matrix work multiplied zeros.
template <int T>
__global__ void __launch_bounds__(256)
    linear_wgrad_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ x, int ldx, int M, int N,
                        int K, int rows_per_split, float *__restrict__ partial, int with_bias) {
  static_assert(T == 128 || T == 64, "tile");
  constexpr int NT = T / 64;                          // 32 x 32 MFMA tiles per wave and dimension
  constexpr int CPR = T / 4, RPP = 256 / CPR, NP = WG_BK / RPP;  // staging: chunks per slab row, rows per pass, passes
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * T * WG_LDT];  // [As0 | As1 | Bs0 | Bs1]
  float(*As)[T * WG_LDT] = reinterpret_cast<float(*)[T * WG_LDT]>(smem);
  float(*Bs)[T * WG_LDT] = reinterpret_cast<float(*)[T * WG_LDT]>(smem + 2 * T * WG_LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int k0 = blockIdx.x * T, n0 = blockIdx.y * T;  // tile of dW [N, K]: rows n, columns k
  const int mb = blockIdx.z * rows_per_split, me = min(M, mb + rows_per_split);

  // staging: a slab is 16 batch rows x T columns of dY (and of X); thread -> (row r, 4-column chunk c4)
  const int sr = tid / CPR, sc = (tid % CPR) * 4;
  float4 pa[NP], pb[NP];
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int m = m0 + sr + RPP * i;
      pa[i] = pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < me) {
        if (n0 + sc < N) pa[i] = *reinterpret_cast<const float4 *>(dy + (size_t)m * lddy + n0 + sc);
        if (k0 + sc < K) pb[i] = *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + k0 + sc);
      }
    }
  };
  auto sstore = [&](int buf) {  // transposed: LDS row = output index (n or k), LDS column = batch row of the slab
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = sr + RPP * i;
      As[buf][(sc + 0) * WG_LDT + r] = pa[i].x, As[buf][(sc + 1) * WG_LDT + r] = pa[i].y;
      As[buf][(sc + 2) * WG_LDT + r] = pa[i].z, As[buf][(sc + 3) * WG_LDT + r] = pa[i].w;
      Bs[buf][(sc + 0) * WG_LDT + r] = pb[i].x, Bs[buf][(sc + 1) * WG_LDT + r] = pb[i].y;
      Bs[buf][(sc + 2) * WG_LDT + r] = pb[i].z, Bs[buf][(sc + 3) * WG_LDT + r] = pb[i].w;
    }
  };

  f32x16 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // bias gradient = column sums of dY: the workgroups of the first k-tile add up the slabs they stage anyway
  const bool do_bias = with_bias && blockIdx.x == 0 && tid < T;
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
    float4 a[NT][2], b[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        a[t][v] = *reinterpret_cast<const float4 *>(&As[buf][(wm * (T / 2) + t * 32 + l31) * WG_LDT + 8 * half + 4 * v]);
        b[t][v] = *reinterpret_cast<const float4 *>(&Bs[buf][(wn * (T / 2) + t * 32 + l31) * WG_LDT + 8 * half + 4 * v]);
      }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
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
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int k = k0 + wn * (T / 2) + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * (T / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < N && k < K) dst[(size_t)n * K + k] = acc[i][j][r];
      }
    }
}

// out[i] = sum_s partial[s * stride + i] in a FIXED order: sixteen partial sums per output element (sum l adds the splits
// l, l + 16, ... in ascending order), added pairwise in a fixed tree (deterministic; one thread per element walked up to
// 1024 dependent loads -- 90 us per layer at the batch sizes of training)
__global__ void __launch_bounds__(1024)
    reduce_partials_kernel(const float *__restrict__ partial, int S, int64_t stride, int64_t n, float *__restrict__ out) {
  // A workgroup owns 64 consecutive elements: lane = element (a split row is read as one 256-byte piece per wave), wave w of
  // the 16 adds the splits w, w + 16, ... in ascending order, and the sixteen sums meet in LDS in the pairwise tree
  // ((0+8)+(4+12)) + ... of the 16-lane form this replaces (same sums, same order, bit for bit; that form had the sixteen
  // splits of ONE element in adjacent lanes: every 4-byte read its own cache line, 1.6 TB/s).
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float acc = 0.0f;
  if (i < n)
    for (int s = w; s < S; s += 16) acc += partial[(size_t)s * stride + i];
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && i < n) {
    float v[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) v[l] = part[l][lane];
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1)  // lane l of the old form ends with v[l] + v[l ^ o] at every level: the same pairs
#pragma unroll
      for (int l = 0; l < o; ++l) v[l] = v[l] + v[l + o];
    out[i] = v[0];
  }
}

// (train_ops.hip: the sparse pool backward adds its splits through the same kernel)
void mpx_reduce_partials_launch(const float *partial, int S, int64_t stride, int64_t n, float *out, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, stream, partial, S, stride, n, out);
}

// dz = dy * act'(y): ReLU -> y > 0; LeakyReLU(0.01) -> y >= 0 ? 1 : 0.01 (sign of the output = sign of the input)
__global__ void __launch_bounds__(256)
    act_backward_kernel(const float *dy, const float *__restrict__ y, int64_t n, int act,  // (dy may be dz: in place)
                        float *dz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float g = dy[i], o = y[i];
  dz[i] = act == MPX_ACT_RELU ? (o > 0.0f ? g : 0.0f) : (act == MPX_ACT_LEAKY ? (o >= 0.0f ? g : 0.01f * g) : g);
}

// ---- backward of GroupNorm(groups) + LeakyReLU(0.01) on [M, C] (mpx_groupnorm_leaky; model.py:386-391) -----------
// xh = (x - mean) * rstd, o = xh * gamma + beta, y = leaky(o).  One wave per (row, group): recomputes the statistics,
// forms do = dy * leaky'(o) and dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)) with dxh = do * gamma, and leaves
// (mean, rstd) in `stats` for the parameter-gradient pass.
__device__ __forceinline__ float gn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(256)
    groupnorm_leaky_dx_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                              const float *__restrict__ beta, const float *__restrict__ dy, int64_t n_rg, int C,
                              int groups, float eps, float *__restrict__ dx, float *__restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rg >= n_rg) return;
  const int gs = C / groups;
  const int64_t row = rg / groups;
  const int g = (int)(rg % groups);
  const int64_t base = row * C + (int64_t)g * gs;
  constexpr int MAXV = 8;  // gs <= 512
  float v[MAXV], s = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < gs ? x[base + c] : 0.0f;
    s += v[i];
  }
  const float mean = gn_wave_sum(s) / (float)gs;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    const float d = c < gs ? v[i] - mean : 0.0f;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(gn_wave_sum(q) / (float)gs + eps);
  float dxh[MAXV], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    dxh[i] = 0.0f;
    if (c < gs) {
      const float xh = (v[i] - mean) * rstd, ga = gamma[g * gs + c];
      const float o = xh * ga + beta[g * gs + c];
      const float d_o = dy[base + c] * (o >= 0.0f ? 1.0f : 0.01f);
      dxh[i] = d_o * ga;
      s1 += dxh[i];
      s2 += dxh[i] * xh;
      v[i] = xh;
    }
  }
  s1 = gn_wave_sum(s1) / (float)gs;
  s2 = gn_wave_sum(s2) / (float)gs;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < gs) dx[base + c] = rstd * (dxh[i] - s1 - v[i] * s2);
  }
  if (lane == 0) stats[2 * rg] = mean, stats[2 * rg + 1] = rstd;
}

// dgamma[c] = sum_rows do * xh, dbeta[c] = sum_rows do (rows in order: deterministic)
__global__ void __launch_bounds__(256)
    groupnorm_leaky_dparam_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                  const float *__restrict__ beta, const float *__restrict__ dy,
                                  const float *__restrict__ stats, int M, int C, int groups,
                                  float *__restrict__ dgamma, float *__restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int g = c / (C / groups);
  const float ga = gamma[c], be = beta[c];
  float a = 0.0f, b = 0.0f;
  for (int m = 0; m < M; ++m) {
    const float mean = stats[2 * ((int64_t)m * groups + g)], rstd = stats[2 * ((int64_t)m * groups + g) + 1];
    const float xh = (x[(int64_t)m * C + c] - mean) * rstd;
    const float d_o = dy[(int64_t)m * C + c] * (xh * ga + be >= 0.0f ? 1.0f : 0.01f);
    a += d_o * xh;
    b += d_o;
  }
  dgamma[c] = a;
  dbeta[c] = b;
}

MPX_EXPORT int mpx_groupnorm_leaky_grad(const float *x, const float *gamma, const float *beta, const float *dy, int M,
                                        int C, int groups, float eps, float *dx, float *dgamma, float *dbeta,
                                        float *stats, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 0 && C >= 1 && groups >= 1 && C % groups == 0, "mpx_groupnorm_leaky_grad: bad size");
  MPX_REQUIRE(C / groups <= 512, "mpx_groupnorm_leaky_grad: group size %d > 512 unsupported", C / groups);
  MPX_REQUIRE(dx && dgamma && dbeta && stats, "mpx_groupnorm_leaky_grad: NULL output");
  if (M == 0) return 0;
  const int64_t n = (int64_t)M * groups;
  hipLaunchKernelGGL(groupnorm_leaky_dx_kernel, dim3(cdiv(n, 4)), dim3(256), 0, mpx_s(stream), x, gamma, beta, dy, n, C,
                     groups, eps, dx, stats);
  hipLaunchKernelGGL(groupnorm_leaky_dparam_kernel, dim3(cdiv(C, 256)), dim3(256), 0, mpx_s(stream), x, gamma, beta, dy,
                     stats, M, C, groups, dgamma, dbeta);
  MPX_LAUNCH_CHECK("mpx_groupnorm_leaky_grad");
}

static int wgrad_tile(int N, int K) { return (N <= 64 && K <= 64) ? 64 : 128; }
static int wgrad_splits(int M, int N, int K) {
  const int T = wgrad_tile(N, K);
  const int tiles = cdiv(N, T) * cdiv(K, T);
  int S = cdiv(1024, tiles);
  const int maxs = cdiv(M, 4 * WG_BK);
  return S < 1 ? 1 : (S > maxs ? (maxs < 1 ? 1 : maxs) : S);
}
MPX_EXPORT int64_t mpx_linear_wgrad_scratch(int M, int N, int K) {
  return (int64_t)wgrad_splits(M, N, K) * ((int64_t)N * K + N);  // floats: weight partials + bias partials
}

void mpx_wgrad_bf16x3_launch(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, int rows_per_split,
                             int S, float *partial, int with_bias, hipStream_t stream);  // dense_bf16.hip
static int wgrad_run(const char *name, bool x3, const float *dy, int lddy, const float *x, int ldx, int M, int N, int K,
                     float *dw, float *db, float *scratch, mpx_stream_t stream);
MPX_EXPORT int mpx_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, float *dw,
                                float *db, float *scratch, mpx_stream_t stream) {
  return wgrad_run("mpx_linear_wgrad", false, dy, lddy, x, ldx, M, N, K, dw, db, scratch, stream);
}
// the same in the split-bf16 arithmetic (three bf16 MFMAs per fp32 product, fp32 accumulate; dense_bf16.hip); layers of
// <= 64 inputs and outputs keep the fp32 kernel (their 64-wide tile)
MPX_EXPORT int mpx_linear_wgrad_bf16x3(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, float *dw,
                                       float *db, float *scratch, mpx_stream_t stream) {
  return wgrad_run("mpx_linear_wgrad_bf16x3", true, dy, lddy, x, ldx, M, N, K, dw, db, scratch, stream);
}
static int wgrad_run(const char *name, bool x3, const float *dy, int lddy, const float *x, int ldx, int M, int N, int K,
                     float *dw, float *db, float *scratch, mpx_stream_t stream) {
  MPX_REQUIRE(M >= 1 && N >= 1 && K >= 1, "%s: bad size", name);
  MPX_REQUIRE(N % 4 == 0 && K % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0,
              "%s: N, K and the leading dimensions must be multiples of 4", name);
  MPX_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0, "%s: operands must be 16-byte aligned", name);
  MPX_REQUIRE(dw && scratch, "%s: NULL output / scratch", name);
  const int64_t per = (int64_t)N * K + N;
  const int S = wgrad_splits(M, N, K), T = wgrad_tile(N, K);
  const int rps = cdiv(cdiv(M, S), WG_BK) * WG_BK;
  MPX_REQUIRE(cdiv(N, T) <= 65535 && S <= 65535, "%s: grid too large", name);
  // ONE split (few rows: the reference's batch of 10 through the dense heads) and dw | db adjacent, as every split's slice
  // is laid out: the kernel writes the gradients themselves.  (The reduction of a single split was a copy on 16 lanes per
  // element -- 134 M threads for the 4096 x 2048 layer: 0.72 of the 5.2 ms of a batch-10 step.)
  const bool direct = S == 1 && (db == nullptr || db == dw + (size_t)N * K);
  if (direct) scratch = dw;
  if (x3 && T == 128)
    mpx_wgrad_bf16x3_launch(dy, lddy, x, ldx, M, N, K, rps, S, scratch, db ? 1 : 0, mpx_s(stream));
  else if (T == 64)
    hipLaunchKernelGGL(linear_wgrad_kernel<64>, dim3(cdiv(K, T), cdiv(N, T), S), dim3(256), 0, mpx_s(stream), dy, lddy, x,
                       ldx, M, N, K, rps, scratch, db ? 1 : 0);
  else
    hipLaunchKernelGGL(linear_wgrad_kernel<128>, dim3(cdiv(K, T), cdiv(N, T), S), dim3(256), 0, mpx_s(stream), dy, lddy, x,
                       ldx, M, N, K, rps, scratch, db ? 1 : 0);
  // dw [N*K] and db [N] are adjacent in every split's slice: one reduction (db lands right behind dw if the caller
  // laid them out that way, else two launches)
  if (direct) {
  } else if (db == dw + (size_t)N * K) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(per, 64)), dim3(1024), 0, mpx_s(stream), scratch, S, per, per, dw);
  } else {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv((int64_t)N * K, 64)), dim3(1024), 0, mpx_s(stream), scratch, S,
                       per, (int64_t)N * K, dw);
    if (db)
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv((int64_t)N, 64)), dim3(1024), 0, mpx_s(stream),
                         scratch + (size_t)N * K, S, per, (int64_t)N, db);
  }
  MPX_LAUNCH_CHECK(name);
}

MPX_EXPORT int mpx_act_backward(const float *dy, const float *y, int64_t n, int act, float *dz, mpx_stream_t stream) {
  MPX_REQUIRE(n >= 0 && act >= 0 && act <= 2, "mpx_act_backward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_backward_kernel, dim3(cdiv(n, 256)), dim3(256), 0, mpx_s(stream), dy, y, n, act, dz);
  MPX_LAUNCH_CHECK("mpx_act_backward");
}
