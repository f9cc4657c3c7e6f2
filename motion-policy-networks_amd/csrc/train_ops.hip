// train_ops.hip -- next row N1: the differentiable pieces of the
// set-abstraction stack that are not plain dense algebra.  The reference trains through
// pointnet2_ops' QueryAndGroup (grouping + its scatter backward) and a max-pool over the nsample axis
// (model.py:366-383 via pointnet2_modules).  Here the grouped tensor is never padded: a neighbourhood
// contributes only its DISTINCT neighbours (ball-query padding repeats the first hit; a max-pool
// and its gradient ignore repeats), packed as rows of one [R, 3+C] matrix with a segment per query.
#include "common.h"

// rows[off[q] + r, :] = [xyz[idx[q,r]] - new_xyz[q] | feat[idx[q,r], :] | 0 ...],  r < max(cnt[q], 1)
// One wave per query.  The (row, column) elements of the query's block are dealt to the lanes as ONE flat sequence (lane
// takes elements lane, lane + 64, ...: a row of 68 floats does not cost two passes of 64 lanes), the neighbour ids of 64
// rows at a time sit in the lanes and are fetched with a shuffle, and four elements per lane are in flight -- a first form
// walked the rows one by one with a dependent id -> row load chain (0.45 ms for the second module at batch 256).
__global__ void __launch_bounds__(256)
    pack_rows_kernel(const float *__restrict__ xyz, int xs, const float *__restrict__ nxyz, int ns,
                     const float *__restrict__ feat, int fs, int C, const int32_t *__restrict__ idx,
                     const int32_t *__restrict__ cnt, const int64_t *__restrict__ off, int64_t Q, int N, int npoint,
                     int nsample, float *__restrict__ rows, int ld) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, W = 3 + C;
  const int64_t b = q / npoint;
  const int n = max(cnt[q], 1);
  const float cx = nxyz[q * ns], cy = nxyz[q * ns + 1], cz = nxyz[q * ns + 2];
  const int32_t *id = idx + q * nsample;
  float *dst = rows + off[q] * ld;
  const int drow = 64 / ld, dch = 64 % ld;  // one step of 64 elements in (row, column)
  for (int rb = 0; rb < n; rb += 64) {  // 64 rows per window: their ids in the lanes
    const int nr = min(64, n - rb);
    const int my_id = id[rb + min(lane, nr - 1)];
    int row = lane / ld, ch = lane % ld;
    const int total = nr * ld;
    for (int e = lane; e - lane < total; e += 256) {  // four elements per lane per trip (uniform trip count)
      float v[4];
      int rr[4], cc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rr[u] = row, cc[u] = ch;
        const bool in = e + 64 * u < total;
        const int64_t p = b * N + __shfl(my_id, in ? row : 0);
        v[u] = 0.0f;
        if (in) {
          if (ch < 3) v[u] = xyz[p * xs + ch] - (ch == 0 ? cx : ch == 1 ? cy : cz);
          else if (ch < W) v[u] = feat[p * fs + (ch - 3)];
        }
        row += drow, ch += dch;
        if (ch >= ld) ch -= ld, ++row;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e + 64 * u < total) dst[(int64_t)(rb + rr[u]) * ld + cc[u]] = v[u];
    }
  }
}

// dfeat[b, idx[q,r], c] += drows[off[q] + r, 3 + c]   (the coordinates carry no gradient: they are data)
// (the same flat dealing of (row, channel) elements; W = floats between gradient rows)
__global__ void __launch_bounds__(256)
    pack_rows_grad_kernel(const float *__restrict__ drows, int C, const int32_t *__restrict__ idx,
                          const int32_t *__restrict__ cnt, const int64_t *__restrict__ off, int64_t Q, int N,
                          int npoint, int nsample, float *__restrict__ dfeat, int fs, int W) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = q / npoint;
  const int n = max(cnt[q], 1);
  const int32_t *id = idx + q * nsample;
  const float *src = drows + off[q] * W + 3;
  const int drow = 64 / C, dch = 64 % C;
  for (int rb = 0; rb < n; rb += 64) {
    const int nr = min(64, n - rb);
    const int my_id = id[rb + min(lane, nr - 1)];
    int row = lane / C, ch = lane % C;
    const int total = nr * C;
    for (int e = lane; e - lane < total; e += 256) {
      float v[4];
      int64_t pp[4];
      int cc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = e + 64 * u < total;
        pp[u] = b * N + __shfl(my_id, in ? row : 0);
        cc[u] = ch;
        v[u] = in ? src[(int64_t)(rb + row) * W + ch] : 0.0f;
        row += drow, ch += dch;
        if (ch >= C) ch -= C, ++row;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e + 64 * u < total) atomicAdd(dfeat + pp[u] * fs + cc[u], v[u]);
    }
  }
}

// out[q,c] = max_r y[off[q]+r, c];  arg[q,c] = first row attaining it (torch.max keeps the first index)
// (four rows are loaded before they are compared in order: one wave per query has nothing else to hide a load behind)
__global__ void __launch_bounds__(256)
    segment_max_kernel(const float *__restrict__ y, int C, const int64_t *__restrict__ off, int64_t Q,
                       float *__restrict__ out, int os, int64_t *__restrict__ arg) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int lane = threadIdx.x & 63;
  const int64_t r0 = off[q], r1 = off[q + 1];
  for (int c = lane; c < C; c += 64) {
    float best = y[r0 * C + c];
    int64_t at = r0;
    int64_t r = r0 + 1;
    for (; r + 4 <= r1; r += 4) {
      const float v0 = y[r * C + c], v1 = y[(r + 1) * C + c], v2 = y[(r + 2) * C + c], v3 = y[(r + 3) * C + c];
      if (v0 > best) best = v0, at = r;
      if (v1 > best) best = v1, at = r + 1;
      if (v2 > best) best = v2, at = r + 2;
      if (v3 > best) best = v3, at = r + 3;
    }
    for (; r < r1; ++r) {
      const float v = y[r * C + c];
      if (v > best) best = v, at = r;
    }
    out[q * os + c] = best;
    arg[q * C + c] = at;
  }
}

// dy[r, c] = (r == arg[q,c]) ? dout[q,c] * act'(out[q,c]) : 0 for every row r of segment q  (the pooled value IS the
// activation output at the arg-max row: the elementwise backward of the layer in front of the pool costs nothing extra
// here; a query's wave writes ALL rows of its segment, so the caller needs no zero-fill pass over the [R, C] gradient)
__global__ void __launch_bounds__(256)
    segment_max_grad_act_kernel(const float *__restrict__ dout, int ds, const int64_t *__restrict__ arg,
                                const float *__restrict__ out, int os, const int64_t *__restrict__ off, int64_t Q, int C,
                                int act, float *__restrict__ dy) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int lane = threadIdx.x & 63;
  const int64_t r0 = off[q], r1 = off[q + 1];
  for (int c = lane; c < C; c += 64) {
    const float g = dout[q * ds + c], o = out[q * os + c];
    const float v = act == MPX_ACT_RELU ? (o > 0.0f ? g : 0.0f) : (act == MPX_ACT_LEAKY ? (o >= 0.0f ? g : 0.01f * g) : g);
    const int64_t at = arg[q * C + c];
    for (int64_t r = r0; r < r1; ++r) dy[r * C + c] = r == at ? v : 0.0f;
  }
}

// dy[arg[q,c], c] = dout[q,c]  (dy zero-filled by the caller; every (q,c) owns a distinct element)
__global__ void __launch_bounds__(256)
    segment_max_grad_kernel(const float *__restrict__ dout, int ds, const int64_t *__restrict__ arg, int64_t total,
                            int C, float *__restrict__ dy) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t q = i / C;
  const int c = (int)(i - q * C);
  dy[arg[i] * C + c] = dout[q * ds + c];
}

// ---- backward of [dense layer + activation + segment max-pool] WITHOUT the [R, C] gradient ------------------------------
// The pool sends a query's gradient to ONE row per channel (its arg-max): the gradient in front of the pool,
// dZ[r, c] = (r == arg[q,c]) * dout[q,c] * act'(out[q,c]), holds Q*C non-zeros in R*C elements (R / Q = 15 - 60 rows per
// query in the set-abstraction modules, 128 in the group-all module).  The dense route writes dZ (2 GB for the second
// module at batch 256), reads it twice and multiplies zeros in both GEMMs of the layer.  Here both products walk the
// non-zeros:
//   dX[r, k] = act_below'(x[r,k]) * sum_{c : arg[q,c] == r} gz[q,c] * W[c,k]      (mpx_pool_dgrad)
//   dW[c, k] = sum_q gz[q,c] * x[arg[q,c], k],   db[c] = sum_q gz[q,c]            (mpx_pool_wgrad)
// fp32 FMAs in an order fixed by the launch (a different summation order than the dense kernels; inside a row of
// mpx_pool_dgrad the channels are added in the order of an LDS counter's ranks: the hardware's lane order, the same
// from run to run -- tests/test_gpu_training.py asserts repeated launches bit-identical).
constexpr int PB_KC = 64;  // columns of a wave's task: one per lane
__device__ __forceinline__ float pool_gz(float g, float o, int act) {
  return act == MPX_ACT_RELU ? (o > 0.0f ? g : 0.0f) : (act == MPX_ACT_LEAKY ? (o >= 0.0f ? g : 0.01f * g) : g);
}
__device__ __forceinline__ float bcast_f(float v, int l) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

// one wave per (query, 64-column chunk).  The query's live channels (gradient != 0) are ordered by their arg-max row with a
// counting sort in LDS; the wave then walks the ordered list with ONE running sum per lane (its column) and stores the row
// (times the activation backward of the layer below) whenever the row changes; rows nobody points at are written as zeros
// on the way.  No read-modify-write on memory: a first form accumulated into an LDS window with ds_add_f32, which runs at
// ~200 cycles per wave instruction here (5.4 ms for the second module at batch 256), and the window's 32 KB left one wave
// per SIMD to hide the loads.  The weight rows and activation inputs of the next 16 list entries are requested while the
// current 16 are consumed.  Segments of more than PD_ROWS rows are walked in windows of PD_ROWS rows.
// LDS (mpx_pool_dgrad sizes it): row counters | per-channel scratch | ordered list.
constexpr int PD_ROWS = 256;
template <bool BELOW>
__global__ void __launch_bounds__(64)
    pool_dgrad_kernel(const float *__restrict__ dout, int ds, const int64_t *__restrict__ arg, const float *__restrict__ out,
                      int os, const int64_t *__restrict__ off, int C, int act, const float *__restrict__ w, int ldw,
                      const float *__restrict__ x, int ldx, int below, int K, float *__restrict__ gx, int ldg) {
  extern __shared__ __attribute__((aligned(16))) int pool_lds[];
  const int Cp = (C + 63) & ~63;
  int *hist = pool_lds;                                      // [PD_ROWS + 64]: entries per row, then first entry of a row
  int *tmp_r = hist + PD_ROWS + 64;                          // [Cp] row of channel c inside the window, -1 = not listed
  int *tmp_k = tmp_r + Cp;                                   // [Cp] its rank among the row's channels
  float *tmp_g = reinterpret_cast<float *>(tmp_k + Cp);      // [Cp] its gradient
  unsigned *ent = reinterpret_cast<unsigned *>(tmp_g + Cp);  // [Cp + 16] ordered list: row << 16 | channel
  float *entg = reinterpret_cast<float *>(ent + Cp + 16);    // [Cp + 16]
  const int64_t q = blockIdx.x;
  const int kc = blockIdx.y * PB_KC, lane = threadIdx.x;
  const int64_t r0 = off[q];
  const int n = (int)(off[q + 1] - r0);
  const bool live = kc + lane < K;
  const int col = kc + (live ? lane : 0);
  const float *wl = w + col;
  for (int base = 0; base < n; base += PD_ROWS) {
    const int nw = min(PD_ROWS, n - base);
    float *grow = gx + (r0 + base) * ldg + col;          // this window's rows, this lane's column
    const float *xrow = BELOW ? x + (r0 + base) * ldx + col : nullptr;
    for (int r = lane; r < nw + 1; r += 64) hist[r] = 0;
    for (int cb0 = 0; cb0 < C; cb0 += 256) {  // count: channel -> (row, rank inside the row); four blocks of loads in flight
      int64_t a[4];
      float dv[4], ov[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = min(cb0 + 64 * u + lane, C - 1);
        a[u] = arg[q * C + cc], dv[u] = dout[q * ds + cc], ov[u] = out[q * os + cc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = cb0 + 64 * u + lane;
        if (cb0 + 64 * u < C) {  // (uniform)
          const int rr = (int)(a[u] - r0) - base;
          const float g = pool_gz(dv[u], ov[u], act);
          const bool in = c < C && rr >= 0 && rr < nw && g != 0.0f;
          int rank = 0;
          if (in) rank = atomicAdd(&hist[rr], 1);
          tmp_r[c] = in ? rr : -1;
          tmp_k[c] = rank;
          tmp_g[c] = g;
        }
      }
    }
    int total;
    {  // exclusive scan of the row counters (4 consecutive rows per lane)
      static_assert(PD_ROWS <= 256, "four rows per lane");
      int cnt[4], sum = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * lane + i;
        cnt[i] = r < nw ? hist[r] : 0;
        sum += cnt[i];
      }
      int inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      total = __builtin_amdgcn_readlane(inc, 63);
      int start = inc - sum;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * lane + i;
        if (r < nw) hist[r] = start;
        start += cnt[i];
      }
    }
    for (int cb = 0; cb < C; cb += 64) {  // scatter into row order
      const int c = cb + lane;
      const int rr = tmp_r[c];
      if (rr >= 0) {
        const int pos = hist[rr] + tmp_k[c];
        ent[pos] = ((unsigned)rr << 16) | (unsigned)c;
        entg[pos] = tmp_g[c];
      }
    }
    if (lane < 16) {  // the tail batch: copies of the last entry with gradient 0 (same row: no store, adds 0)
      ent[total + lane] = total > 0 ? ent[total - 1] : 0u;
      entg[total + lane] = 0.0f;
    }
    auto zero_rows = [&](int ra, int rb) __attribute__((always_inline)) {
      for (int r = ra; r < rb; ++r)
        if (live) grow[(int64_t)r * ldg] = 0.0f;
    };
    if (total == 0) {
      zero_rows(0, nw);
      continue;
    }
    // ---- the walk: batches of 16 entries; lanes 0..15 hold the batch's (row | channel, gradient) words
    auto load_ent = [&](int e0, unsigned &ev, float &gv) __attribute__((always_inline)) {
      const int e = min(e0 + (lane & 15), total + 15);
      ev = ent[e], gv = entg[e];
    };
    auto load_wx = [&](unsigned ev, float(&wv)[16], float(&xv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned word = (unsigned)__builtin_amdgcn_readlane((int)ev, e);
        wv[e] = wl[(size_t)(word & 0xFFFFu) * ldw];
        xv[e] = BELOW ? xrow[(int64_t)(word >> 16) * ldx] : 0.0f;
      }
    };
    float acc = 0.0f, xcur = 0.0f;
    int cur = 0;
    auto consume = [&](unsigned ev, float gv, const float(&wv)[16], const float(&xv)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rs = (int)((unsigned)__builtin_amdgcn_readlane((int)ev, e) >> 16);
        if (rs != cur) {  // (uniform) the row is complete: store it, zeros for the rows nobody points at
          if (live) grow[(int64_t)cur * ldg] = BELOW ? pool_gz(acc, xcur, below) : acc;
          zero_rows(cur + 1, rs);
          acc = 0.0f, cur = rs, xcur = xv[e];
        }
        acc = fmaf(bcast_f(gv, e), wv[e], acc);
      }
    };
    unsigned ea, eb;
    float ga, gb, wa[16], wb[16], xa[16], xb[16];
    load_ent(0, ea, ga);
    cur = (int)((unsigned)__builtin_amdgcn_readlane((int)ea, 0) >> 16);
    load_wx(ea, wa, xa);
    load_ent(16, eb, gb);
    zero_rows(0, cur);
    xcur = xa[0];
#pragma unroll 1
    for (int e0 = 0; e0 < total; e0 += 32) {  // (two batches per trip: the buffers alternate without a copy)
      load_wx(eb, wb, xb);
      consume(ea, ga, wa, xa);
      load_ent(e0 + 32, ea, ga);
      load_wx(ea, wa, xa);
      if (e0 + 16 < total) consume(eb, gb, wb, xb);
      load_ent(e0 + 48, eb, gb);
    }
    if (live) grow[(int64_t)cur * ldg] = BELOW ? pool_gz(acc, xcur, below) : acc;
    zero_rows(cur + 1, nw);
  }
}

// one wave per (64 channels, 64 columns, split of the queries): lane = column; the wave's 64 channels accumulate in 64
// registers; per query the arg-max rows of 16 channels are requested ahead of their use.  Partial tiles [S][C*K + C] (the
// bias gradient behind the weight gradient, as mpx_linear_wgrad lays its splits out), added in a fixed order afterwards.
__global__ void __launch_bounds__(64)
    pool_wgrad_kernel(const float *__restrict__ dout, int ds, const int64_t *__restrict__ arg, const float *__restrict__ out,
                      int os, int64_t Q, int C, int act, const float *__restrict__ x, int ldx, int K, int64_t q_per_split,
                      float *__restrict__ partial, int with_bias) {
  const int cb = blockIdx.x * 64, kc = blockIdx.y * PB_KC, lane = threadIdx.x;
  const int64_t q0 = (int64_t)blockIdx.z * q_per_split, q1 = min(Q, q0 + q_per_split);
  const int c = cb + lane;
  const bool live = kc + lane < K;
  const float *xl = x + kc + (live ? lane : 0);
  float acc[64], accb = 0.0f;
#pragma unroll
  for (int e = 0; e < 64; ++e) acc[e] = 0.0f;
  for (int64_t q = q0; q < q1; ++q) {
    float gz = 0.0f;
    int row = 0;
    if (c < C) {
      gz = pool_gz(dout[q * ds + c], out[q * os + c], act);
      row = (int)arg[q * C + c];
    }
    accb += gz;
#pragma unroll
    for (int e0 = 0; e0 < 64; e0 += 16) {
      float xv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) xv[e] = xl[(size_t)__builtin_amdgcn_readlane(row, e0 + e) * ldx];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e0 + e] = fmaf(bcast_f(gz, e0 + e), xv[e], acc[e0 + e]);
    }
  }
  float *p = partial + (size_t)blockIdx.z * ((size_t)C * K + C);
  if (live) {
#pragma unroll
    for (int e = 0; e < 64; ++e)
      if (cb + e < C) p[(size_t)(cb + e) * K + kc + lane] = acc[e];
  }
  if (with_bias && blockIdx.y == 0 && c < C) p[(size_t)C * K + c] = accb;
}

static int64_t pool_wgrad_splits(int64_t Q, int C, int K) {
  const int64_t blocks = (int64_t)cdiv(C, 64) * cdiv(K, PB_KC);
  int64_t S = cdiv((int64_t)4096, blocks);
  const int64_t maxs = cdiv(Q, (int64_t)8);  // >= 8 queries per split
  if (S > maxs) S = maxs;
  return S < 1 ? 1 : (S > 65535 ? 65535 : S);
}
void mpx_reduce_partials_launch(const float *partial, int S, int64_t stride, int64_t n, float *out, hipStream_t stream);  // dense_grad.hip

MPX_EXPORT int64_t mpx_pool_wgrad_scratch(int64_t Q, int C, int K) {
  return pool_wgrad_splits(Q, C, K) * ((int64_t)C * K + C);  // floats
}

MPX_EXPORT int mpx_pool_wgrad(const float *grad_out, int grad_stride, const int64_t *arg, const float *out, int out_stride,
                              int64_t Q, int C, int act, const float *x, int ldx, int K, float *dw, float *db,
                              float *scratch, mpx_stream_t stream) {
  MPX_REQUIRE(Q >= 1 && C >= 1 && K >= 1 && grad_stride >= C && out_stride >= C && ldx >= K, "mpx_pool_wgrad: bad size");
  MPX_REQUIRE(act >= 0 && act <= 2, "mpx_pool_wgrad: bad activation");
  MPX_REQUIRE(grad_out && arg && out && x && dw && scratch, "mpx_pool_wgrad: NULL operand / scratch");
  MPX_REQUIRE(db == nullptr || db == dw + (size_t)C * K, "mpx_pool_wgrad: db must follow dw (one reduction over dw | db)");
  const int64_t S = pool_wgrad_splits(Q, C, K), per = (int64_t)C * K + C;
  hipLaunchKernelGGL(pool_wgrad_kernel, dim3(cdiv(C, 64), cdiv(K, PB_KC), (unsigned)S), dim3(64), 0, mpx_s(stream), grad_out,
                     grad_stride, arg, out, out_stride, Q, C, act, x, ldx, K, cdiv(Q, S), scratch, db ? 1 : 0);
  mpx_reduce_partials_launch(scratch, (int)S, per, db ? per : (int64_t)C * K, dw, mpx_s(stream));
  MPX_LAUNCH_CHECK("mpx_pool_wgrad");
}

MPX_EXPORT int mpx_pool_dgrad(const float *grad_out, int grad_stride, const int64_t *arg, const float *out, int out_stride,
                              const int64_t *offsets, int64_t Q, int C, int act, const float *w, int ldw, const float *x,
                              int ldx, int below, int K, int max_rows, float *gx, int ldg, mpx_stream_t stream) {
  MPX_REQUIRE(Q >= 1 && C >= 1 && K >= 1 && grad_stride >= C && out_stride >= C && ldw >= K && ldg >= K,
              "mpx_pool_dgrad: bad size");
  MPX_REQUIRE(act >= 0 && act <= 2 && below >= 0 && below <= 2, "mpx_pool_dgrad: bad activation");
  MPX_REQUIRE(grad_out && arg && out && offsets && w && gx && (below == 0 || (x && ldx >= K)),
              "mpx_pool_dgrad: NULL operand (x is required with an activation below)");
  MPX_REQUIRE(Q <= 0x7FFFFFFF && cdiv(K, PB_KC) <= 65535, "mpx_pool_dgrad: grid too large");
  MPX_REQUIRE(C <= 65535, "mpx_pool_dgrad: more than 65535 channels");
  (void)max_rows;  // (a hint kept in the signature: the kernel holds no row window any more)
  const int Cp = (C + 63) & ~63;
  const size_t lds = (size_t)(PD_ROWS + 64) * 4 + (size_t)Cp * 12 + (size_t)(Cp + 16) * 8;
  MPX_REQUIRE(lds <= 64 * 1024, "mpx_pool_dgrad: %d channels need %zu bytes of LDS", C, lds);
  if (below)
    hipLaunchKernelGGL(pool_dgrad_kernel<true>, dim3((unsigned)Q, cdiv(K, PB_KC)), dim3(64), lds, mpx_s(stream), grad_out,
                       grad_stride, arg, out, out_stride, offsets, C, act, w, ldw, x, ldx, below, K, gx, ldg);
  else
    hipLaunchKernelGGL(pool_dgrad_kernel<false>, dim3((unsigned)Q, cdiv(K, PB_KC)), dim3(64), lds, mpx_s(stream), grad_out,
                       grad_stride, arg, out, out_stride, offsets, C, act, w, ldw, x, ldx, below, K, gx, ldg);
  MPX_LAUNCH_CHECK("mpx_pool_dgrad");
}

MPX_EXPORT int mpx_pack_rows(const float *xyz, int xyz_stride, const float *new_xyz, int new_stride,
                             const float *feat, int feat_stride, int C, const int32_t *idx, const int32_t *cnt,
                             const int64_t *offsets, int B, int N, int npoint, int nsample, float *rows,
                             mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N > 0 && npoint > 0 && nsample > 0 && C >= 0, "mpx_pack_rows: bad size");
  MPX_REQUIRE(xyz_stride >= 3 && new_stride >= 3 && (C == 0 || feat_stride >= C), "mpx_pack_rows: bad stride");
  const int64_t Q = (int64_t)B * npoint;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), xyz, xyz_stride, new_xyz,
                     new_stride, feat, feat_stride, C, idx, cnt, offsets, Q, N, npoint, nsample, rows, 3 + C);
  MPX_LAUNCH_CHECK("mpx_pack_rows");
}
// the same with rows `ld` >= 3 + C floats apart, the columns behind 3 + C zero-filled (ld a multiple of 4: the rows feed the
// GEMMs without a padding copy)
MPX_EXPORT int mpx_pack_rows_ld(const float *xyz, int xyz_stride, const float *new_xyz, int new_stride, const float *feat,
                                int feat_stride, int C, const int32_t *idx, const int32_t *cnt, const int64_t *offsets, int B,
                                int N, int npoint, int nsample, float *rows, int ld, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N > 0 && npoint > 0 && nsample > 0 && C >= 0 && ld >= 3 + C, "mpx_pack_rows_ld: bad size");
  MPX_REQUIRE(xyz_stride >= 3 && new_stride >= 3 && (C == 0 || feat_stride >= C), "mpx_pack_rows_ld: bad stride");
  const int64_t Q = (int64_t)B * npoint;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), xyz, xyz_stride, new_xyz,
                     new_stride, feat, feat_stride, C, idx, cnt, offsets, Q, N, npoint, nsample, rows, ld);
  MPX_LAUNCH_CHECK("mpx_pack_rows_ld");
}

MPX_EXPORT int mpx_pack_rows_grad(const float *grad_rows, int C, const int32_t *idx, const int32_t *cnt,
                                  const int64_t *offsets, int B, int N, int npoint, int nsample, float *grad_feat,
                                  int feat_stride, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N > 0 && npoint > 0 && nsample > 0 && C > 0 && feat_stride >= C, "mpx_pack_rows_grad: bad size");
  const int64_t Q = (int64_t)B * npoint;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(pack_rows_grad_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), grad_rows, C, idx, cnt,
                     offsets, Q, N, npoint, nsample, grad_feat, feat_stride, 3 + C);
  MPX_LAUNCH_CHECK("mpx_pack_rows_grad");
}
// (gradient rows `ld` >= 3 + C floats apart)
MPX_EXPORT int mpx_pack_rows_grad_ld(const float *grad_rows, int ld, int C, const int32_t *idx, const int32_t *cnt,
                                     const int64_t *offsets, int B, int N, int npoint, int nsample, float *grad_feat,
                                     int feat_stride, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N > 0 && npoint > 0 && nsample > 0 && C > 0 && feat_stride >= C && ld >= 3 + C,
              "mpx_pack_rows_grad_ld: bad size");
  const int64_t Q = (int64_t)B * npoint;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(pack_rows_grad_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), grad_rows, C, idx, cnt,
                     offsets, Q, N, npoint, nsample, grad_feat, feat_stride, ld);
  MPX_LAUNCH_CHECK("mpx_pack_rows_grad_ld");
}

MPX_EXPORT int mpx_segment_max(const float *y, int C, const int64_t *offsets, int64_t Q, float *out, int out_stride,
                               int64_t *arg, mpx_stream_t stream) {
  MPX_REQUIRE(Q >= 0 && C > 0 && out_stride >= C, "mpx_segment_max: bad size");
  if (Q == 0) return 0;
  hipLaunchKernelGGL(segment_max_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), y, C, offsets, Q, out,
                     out_stride, arg);
  MPX_LAUNCH_CHECK("mpx_segment_max");
}

MPX_EXPORT int mpx_segment_max_grad(const float *grad_out, int grad_stride, const int64_t *arg, int64_t Q, int C,
                                    float *grad_y, mpx_stream_t stream) {
  MPX_REQUIRE(Q >= 0 && C > 0 && grad_stride >= C, "mpx_segment_max_grad: bad size");
  if (Q == 0) return 0;
  hipLaunchKernelGGL(segment_max_grad_kernel, dim3(cdiv(Q * C, 256)), dim3(256), 0, mpx_s(stream), grad_out,
                     grad_stride, arg, Q * C, C, grad_y);
  MPX_LAUNCH_CHECK("mpx_segment_max_grad");
}

MPX_EXPORT int mpx_segment_max_grad_act(const float *grad_out, int grad_stride, const int64_t *arg, const float *out,
                                        int out_stride, const int64_t *offsets, int64_t Q, int C, int act, float *grad_y,
                                        mpx_stream_t stream) {
  MPX_REQUIRE(Q >= 0 && C > 0 && grad_stride >= C && out_stride >= C, "mpx_segment_max_grad_act: bad size");
  MPX_REQUIRE(act >= 0 && act <= 2 && out != nullptr && offsets != nullptr,
              "mpx_segment_max_grad_act: bad activation / missing pooled rows or offsets");
  if (Q == 0) return 0;
  hipLaunchKernelGGL(segment_max_grad_act_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), grad_out, grad_stride, arg,
                     out, out_stride, offsets, Q, C, act, grad_y);
  MPX_LAUNCH_CHECK("mpx_segment_max_grad_act");
}
