// train_ops.hip -- next row N1: the differentiable pieces of the
// set-abstraction stack that are not plain dense algebra.  The reference trains through
// pointnet2_ops' QueryAndGroup (grouping + its scatter backward) and a max-pool over the nsample axis
// (model.py:366-383 via pointnet2_modules).  Here the grouped tensor is never padded: a neighbourhood
// contributes only its DISTINCT neighbours (ball-query padding repeats the first hit; a max-pool
// and its gradient ignore repeats), packed as rows of one [R, 3+C] matrix with a segment per query.
#include "common.h"

// rows[off[q] + r, :] = [xyz[idx[q,r]] - new_xyz[q] | feat[idx[q,r], :]],  r < max(cnt[q], 1)
__global__ void __launch_bounds__(256)
    pack_rows_kernel(const float *__restrict__ xyz, int xs, const float *__restrict__ nxyz, int ns,
                     const float *__restrict__ feat, int fs, int C, const int32_t *__restrict__ idx,
                     const int32_t *__restrict__ cnt, const int64_t *__restrict__ off, int64_t Q, int N, int npoint,
                     int nsample, float *__restrict__ rows) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, W = 3 + C;
  const int64_t b = q / npoint;
  const int n = max(cnt[q], 1);
  const float cx = nxyz[q * ns], cy = nxyz[q * ns + 1], cz = nxyz[q * ns + 2];
  const int32_t *id = idx + q * nsample;
  float *dst = rows + off[q] * W;
  for (int r = 0; r < n; ++r) {
    const int64_t p = b * N + id[r];
    for (int ch = lane; ch < W; ch += 64) {
      float v;
      if (ch < 3) v = xyz[p * xs + ch] - (ch == 0 ? cx : ch == 1 ? cy : cz);
      else v = feat[p * fs + (ch - 3)];
      dst[(int64_t)r * W + ch] = v;
    }
  }
}

// dfeat[b, idx[q,r], c] += drows[off[q] + r, 3 + c]   (the coordinates carry no gradient: they are data)
__global__ void __launch_bounds__(256)
    pack_rows_grad_kernel(const float *__restrict__ drows, int C, const int32_t *__restrict__ idx,
                          const int32_t *__restrict__ cnt, const int64_t *__restrict__ off, int64_t Q, int N,
                          int npoint, int nsample, float *__restrict__ dfeat, int fs) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int lane = threadIdx.x & 63, W = 3 + C;
  const int64_t b = q / npoint;
  const int n = max(cnt[q], 1);
  const int32_t *id = idx + q * nsample;
  const float *src = drows + off[q] * W;
  for (int r = 0; r < n; ++r) {
    const int64_t p = b * N + id[r];
    for (int c = lane; c < C; c += 64) atomicAdd(dfeat + p * fs + c, src[(int64_t)r * W + 3 + c]);
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

MPX_EXPORT int mpx_pack_rows(const float *xyz, int xyz_stride, const float *new_xyz, int new_stride,
                             const float *feat, int feat_stride, int C, const int32_t *idx, const int32_t *cnt,
                             const int64_t *offsets, int B, int N, int npoint, int nsample, float *rows,
                             mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N > 0 && npoint > 0 && nsample > 0 && C >= 0, "mpx_pack_rows: bad size");
  MPX_REQUIRE(xyz_stride >= 3 && new_stride >= 3 && (C == 0 || feat_stride >= C), "mpx_pack_rows: bad stride");
  const int64_t Q = (int64_t)B * npoint;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), xyz, xyz_stride, new_xyz,
                     new_stride, feat, feat_stride, C, idx, cnt, offsets, Q, N, npoint, nsample, rows);
  MPX_LAUNCH_CHECK("mpx_pack_rows");
}

MPX_EXPORT int mpx_pack_rows_grad(const float *grad_rows, int C, const int32_t *idx, const int32_t *cnt,
                                  const int64_t *offsets, int B, int N, int npoint, int nsample, float *grad_feat,
                                  int feat_stride, mpx_stream_t stream) {
  MPX_REQUIRE(B >= 0 && N > 0 && npoint > 0 && nsample > 0 && C > 0 && feat_stride >= C, "mpx_pack_rows_grad: bad size");
  const int64_t Q = (int64_t)B * npoint;
  if (Q == 0) return 0;
  hipLaunchKernelGGL(pack_rows_grad_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, mpx_s(stream), grad_rows, C, idx, cnt,
                     offsets, Q, N, npoint, nsample, grad_feat, feat_stride);
  MPX_LAUNCH_CHECK("mpx_pack_rows_grad");
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
