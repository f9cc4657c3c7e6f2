// policy.hip -- the whole policy forward behind ONE C-ABI call (MotionPolicyNetwork.forward, model.py:75-91):
// a C / C++ caller (a native planning node) hands over the slab, the joint configuration and a workspace and gets
// the displacement back.  Host-side orchestration only: every stage is one of this library's kernels, launched in
// the order and with the shapes of mpinets_amd/model.py (whose output it reproduces bit for bit), on the caller's
// stream (plus an internal second one for small batches), with no allocation and no synchronisation.
//
// Reference: MPiNetsPointNet (model.py:360-426: SA(512, r 0.05, 128, [1,64,64,64]) -> SA(128, r 0.3, 128,
// [64,128,128,256]) -> group-all [256,512,512,1024] -> 1024-4096-GN-2048-GN-2048), feature_encoder
// (model.py:47-57), decoder (model.py:58-66).
#include "common.h"

#include <mutex>

namespace {

constexpr int NP1 = 512, NP2 = 128, NS = 128;     // samples per module, neighbours per ball
constexpr float R1 = 0.05f, R2 = 0.3f;            // ball radii (model.py:366-381)
constexpr int C1 = 64, F1 = C1 + 4;               // SA1 output channels; its row [f1 | xyz1 | 0]
constexpr int C2 = 256, K3 = 272;                 // SA2 output channels; group-all input row [xyz2 | f2 | 0 x 13]
constexpr int H3 = 512, C3 = 1024;                // group-all hidden / output width
constexpr int ENC = 2048, QF = 64, CAT = ENC + QF;

// rows[i, col0 : col0 + ncols] = src[i, :ncols], rows[i, col0 + ncols : col0 + ncols + nzero] = 0
__global__ void __launch_bounds__(256)
    append_columns_kernel(const float *__restrict__ src, int src_stride, int ncols, int nzero, int64_t n, float *__restrict__ rows,
                          int row_stride, int col0) {
  const int w = ncols + nzero;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * w) return;
  const int64_t i = e / w;
  const int j = (int)(e - i * w);
  rows[i * row_stride + col0 + j] = j < ncols ? src[i * src_stride + j] : 0.0f;
}

// zero columns [3 + C2, K3) of the group-all input rows (the sampling kernel writes xyz2, SA2 writes f2) and column
// 3, which the per-query first-layer GEMM reads -- times a zero weight -- before SA2 has written it
__global__ void __launch_bounds__(256) zero_tail_kernel(float *__restrict__ rows, int64_t n) {
  constexpr int TAIL = K3 - 3 - C2 + 1;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * TAIL) return;
  const int j = (int)(i % TAIL);
  rows[(i / TAIL) * K3 + (j == 0 ? 3 : 3 + C2 + j - 1)] = 0.0f;
}

// q [B,7] -> [B,8] (K of the first joint-encoder layer padded to a multiple of 4)
__global__ void __launch_bounds__(256) pad_q_kernel(const float *__restrict__ q, int B, float *__restrict__ q8) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 8) return;
  const int b = i >> 3, j = i & 7;
  q8[i] = j < 7 ? q[b * 7 + j] : 0.0f;
}

struct Carver {  // hands out 256-byte aligned pieces of the workspace (or only counts, when base == nullptr)
  char *base;
  int64_t used = 0;
  template <class T>
  T *take(int64_t n) {
    T *p = base ? reinterpret_cast<T *>(base + used) : nullptr;
    used += (n * (int64_t)sizeof(T) + 255) / 256 * 256;
    return p;
  }
};

struct Buffers {
  int32_t *idx1, *nbr1, *cnt1, *idx2, *nbr2, *cnt2;
  float *xyz1, *f1, *sa3_in, *pre, *ctr, *h_a, *h_b, *pooled, *fc_a, *fc_b, *cat, *q8, *s_a, *s_b, *q_a, *q_b;
  void *splitk;
  int64_t splitk_bytes;
};

int64_t max_i64(int64_t a, int64_t b) { return a > b ? a : b; }

// Buffers whose lifetimes do not overlap share memory (all on the caller's stream, in launch order):
//   region A: nbr1 (SA1's neighbour lists, dead after SA1's MLP) -> pre (SA2's per-point first-layer rows, dead after
//             SA2's MLP) -> h_a (group-all hidden rows);
//   region B: [f1 | ctr | nbr2] (dead after SA2's MLP) -> h_b.
// 0.74 MB per environment instead of 1.5 MB (6 GB instead of 12 GB at 8192 environments).
int64_t carve(char *base, int B, int N, Buffers &bu) {
  Carver c{base};
  const int64_t b = B;
  bu.idx1 = c.take<int32_t>(b * NP1);
  bu.xyz1 = c.take<float>(b * NP1 * 3);
  bu.cnt1 = c.take<int32_t>(b * NP1);
  bu.sa3_in = c.take<float>(b * NP2 * K3);
  bu.idx2 = c.take<int32_t>(b * NP2);
  bu.cnt2 = c.take<int32_t>(b * NP2);
  {  // region A
    const int64_t start = c.used;
    bu.nbr1 = c.take<int32_t>(b * NP1 * NS);
    const int64_t end_nbr1 = c.used;
    c.used = start;
    bu.pre = c.take<float>(b * NP1 * 128);
    const int64_t end_pre = c.used;
    c.used = start;
    bu.h_a = c.take<float>(b * NP2 * (B <= 8 ? C3 : H3));  // (a handful of problems: also the unpooled last layer)
    c.used = max_i64(max_i64(end_nbr1, end_pre), c.used);
  }
  {  // region B
    const int64_t start = c.used;
    bu.f1 = c.take<float>(b * NP1 * F1);
    bu.ctr = c.take<float>(b * NP2 * 128);
    bu.nbr2 = c.take<int32_t>(b * NP2 * NS);
    const int64_t end_inputs = c.used;
    c.used = start;
    bu.h_b = c.take<float>(b * NP2 * H3);
    c.used = max_i64(end_inputs, c.used);
  }
  bu.pooled = c.take<float>(b * C3);
  bu.fc_a = c.take<float>(b * 4096);
  bu.fc_b = c.take<float>(b * 2048);
  bu.cat = c.take<float>(b * CAT);
  bu.q8 = c.take<float>(b * 8);
  bu.s_a = c.take<float>(b * 512);
  bu.s_b = c.take<float>(b * 512);
  bu.q_a = c.take<float>(b * 128);  // (the joint encoder may run beside the point-cloud chain: its own buffers)
  bu.q_b = c.take<float>(b * 128);
  // one split-K area, sized for the hungriest layer of this batch size
  const int shapes[][3] = {{B * NP2, H3, K3}, {B * NP2, H3, H3}, {B * NP2, C3, H3}, {B, 4096, C3}, {B, 2048, 4096},
                           {B, 2048, 2048}, {B, 512, CAT}, {B, 256, 512}, {B, 128, 256}};
  int64_t need = 0;
  for (const auto &s : shapes) need = max_i64(need, mpx_linear_workspace(s[0], s[1], s[2]));
  bu.splitk_bytes = need;
  bu.splitk = need ? c.take<char>(need) : nullptr;
  (void)N;
  return c.used;
}

// Small batches cannot fill the chip with any single kernel (FPS is one workgroup per problem), so two independent
// branches -- SA2's sampling + ball query and the joint-angle encoder -- are issued on a second stream beside SA1's
// ball query + grouped MLP, forked and joined with events (the same split as model.py's; a fork / join like this is
// hipGraph-capturable).  One stream + two events per device, created on first use.  A caller holds the device's
// Side EXCLUSIVELY from its fork record to its join wait (`busy`): two host threads driving the same device from
// different streams can therefore never re-record each other's fork / join events between a record and the wait that
// consumes it (a stream wait captures the event's state at the time of the call, so re-use after the join wait has
// been enqueued is safe; the side stream itself is in order).
constexpr int OVERLAP_MAX_BATCH = 512;
constexpr int SA3_CHAIN_MIN_BATCH = 256;  // one workgroup per problem: the fused group-all chain wants a full chip (= model.py)
struct Side {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  std::mutex busy;
};
Side *side_of_current_device() {
  static std::mutex mu;
  static Side per_device[64];
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  Side &s = per_device[d];
  if (!s.stream) {
    hipStream_t st;
    hipEvent_t f, j;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&j, hipEventDisableTiming) != hipSuccess)
      return nullptr;
    s.stream = st, s.fork = f, s.join = j;
  }
  return &s;
}

}  // namespace

#define MPX_TRY(call)     \
  do {                    \
    const int rc_ = (call); \
    if (rc_ != 0) return rc_; \
  } while (0)

MPX_EXPORT int mpx_append_columns(const float *src, int src_stride, int ncols, int nzero, int64_t n, float *rows, int row_stride,
                                  int col0, mpx_stream_t stream) {
  MPX_REQUIRE(ncols >= 0 && nzero >= 0 && ncols + nzero >= 1 && n >= 0, "mpx_append_columns: bad size");
  MPX_REQUIRE(src_stride >= ncols && col0 >= 0 && row_stride >= col0 + ncols + nzero, "mpx_append_columns: bad stride");
  if (n == 0) return 0;
  hipLaunchKernelGGL(append_columns_kernel, dim3(cdiv(n * (ncols + nzero), 256)), dim3(256), 0, mpx_s(stream), src, src_stride,
                     ncols, nzero, n, rows, row_stride, col0);
  MPX_LAUNCH_CHECK("mpx_append_columns");
}

MPX_EXPORT int64_t mpx_policy_workspace(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  Buffers bu;
  return carve(nullptr, B, N, bu);
}

MPX_EXPORT int mpx_policy_forward(const mpx_policy_weights *w, const float *xyz, int N, const float *q, int B, float *dq,
                                  void *workspace, int64_t workspace_bytes, mpx_stream_t stream) {
  MPX_REQUIRE(w && xyz && q && dq, "mpx_policy_forward: NULL operand");
  MPX_REQUIRE(B >= 0, "mpx_policy_forward: negative batch");  // (any size: the batched launchers walk slabs)
  MPX_REQUIRE(N >= NP1 && N <= 8192, "mpx_policy_forward: N = %d outside [%d, 8192]", N, NP1);
  if (B == 0) return 0;
  MPX_REQUIRE(workspace && (((uintptr_t)workspace) & 255) == 0, "mpx_policy_forward: workspace must be 256-byte aligned");
  Buffers bu;
  const int64_t need = carve(static_cast<char *>(workspace), B, N, bu);
  MPX_REQUIRE(workspace_bytes >= need, "mpx_policy_forward: workspace of %lld bytes, mpx_policy_workspace asks for %lld",
              (long long)workspace_bytes, (long long)need);
  hipStream_t st = mpx_s(stream);
  auto lin = [&](const float *x, int ldx, const float *wt, const float *b, int M, int Nn, int K, int act, float *y, int ldy) {
    return mpx_linear_ws(x, ldx, wt, b, M, Nn, K, act, y, ldy, bu.splitk, bu.splitk_bytes, stream);
  };

  // the two branches that need only xyz1 / q (second stream for small batches, in line otherwise)
  auto sample_sa2_and_encode_q = [&](mpx_stream_t s2) -> int {
    hipLaunchKernelGGL(zero_tail_kernel, dim3(cdiv((int64_t)B * NP2 * (K3 - 3 - C2 + 1), 256)), dim3(256), 0, mpx_s(s2), bu.sa3_in,
                       (int64_t)B * NP2);
    MPX_TRY(mpx_fps(bu.xyz1, B, NP1, 3, NP2, bu.idx2, bu.sa3_in, K3, s2));
    MPX_TRY(mpx_ball_query_hits(bu.sa3_in, K3, bu.xyz1, 3, B, NP1, NP2, R2, NS, bu.nbr2, bu.cnt2, s2));  // (hit slots only)
    // joint encoder 7 -> 32 -> 64 -> 128 -> 128 -> 64, into the right part of the decoder's input rows
    // (no layer here has K >= 256: none of them touches the split-K area the other stream may be using)
    hipLaunchKernelGGL(pad_q_kernel, dim3(cdiv((int64_t)B * 8, 256)), dim3(256), 0, mpx_s(s2), q, B, bu.q8);
    MPX_TRY(mpx_linear(bu.q8, 8, w->qe_w[0], w->qe_b[0], B, 32, 8, MPX_ACT_LEAKY, bu.q_a, 32, s2));
    MPX_TRY(mpx_linear(bu.q_a, 32, w->qe_w[1], w->qe_b[1], B, 64, 32, MPX_ACT_LEAKY, bu.q_b, 64, s2));
    MPX_TRY(mpx_linear(bu.q_b, 64, w->qe_w[2], w->qe_b[2], B, 128, 64, MPX_ACT_LEAKY, bu.q_a, 128, s2));
    MPX_TRY(mpx_linear(bu.q_a, 128, w->qe_w[3], w->qe_b[3], B, 128, 128, MPX_ACT_LEAKY, bu.q_b, 128, s2));
    MPX_TRY(mpx_linear(bu.q_b, 128, w->qe_w[4], w->qe_b[4], B, QF, 128, MPX_ACT_NONE, bu.cat + ENC, CAT, s2));
    return 0;
  };
  auto module_sa1 = [&]() -> int {  // neighbours within 5 cm, grouped MLP over [p - c ; label] rows
    MPX_TRY(mpx_ball_query_hits(bu.xyz1, 3, xyz, 4, B, N, NP1, R1, NS, bu.nbr1, bu.cnt1, stream));
    // (its rows come out as [f1 | xyz1 | 0]: the operand of SA2's per-point first-layer GEMM)
    MPX_TRY(mpx_sa_mlp(xyz, 4, bu.xyz1, 3, xyz + 3, 4, 1, bu.nbr1, bu.cnt1, B, N, NP1, NS, w->sa1_pack, 64, 64, C1, bu.f1,
                       F1, 1, stream));
    return 0;
  };

  // ---- SA1 (sample 512) and, beside it, SA2's sampling (128 of the 512, neighbours within 30 cm) ---------------
  MPX_TRY(mpx_fps(xyz, B, N, 4, NP1, bu.idx1, bu.xyz1, 3, stream));
  Side *side = B <= OVERLAP_MAX_BATCH ? side_of_current_device() : nullptr;
  if (side) {
    std::lock_guard<std::mutex> exclusive(side->busy);  // fork .. join of one call at a time per device
    hipError_t e = hipEventRecord(side->fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side->stream, side->fork, 0);
    MPX_REQUIRE(e == hipSuccess, "mpx_policy_forward: stream fork failed: %s", hipGetErrorString(e));
    const int rc_side = sample_sa2_and_encode_q(reinterpret_cast<mpx_stream_t>(side->stream));
    const int rc_main = rc_side == 0 ? module_sa1() : 0;
    e = hipEventRecord(side->join, side->stream);  // (always joined, also on an error path)
    if (e == hipSuccess) e = hipStreamWaitEvent(st, side->join, 0);
    if (rc_side != 0) return rc_side;
    if (rc_main != 0) return rc_main;
    MPX_REQUIRE(e == hipSuccess, "mpx_policy_forward: stream join failed: %s", hipGetErrorString(e));
  } else {
    MPX_TRY(module_sa1());
    MPX_TRY(sample_sa2_and_encode_q(stream));
  }
  // ---- SA2: first layer per point / per query, layers 2-3 + max-pool fused ------------------------------------
  MPX_TRY(lin(bu.f1, F1, w->sa2_wpoint, nullptr, B * NP1, 128, F1, MPX_ACT_NONE, bu.pre, 128));
  MPX_TRY(lin(bu.sa3_in, K3, w->sa2_wcentre, w->sa2_nb1, B * NP2, 128, 4, MPX_ACT_NONE, bu.ctr, 128));
  MPX_TRY(mpx_sa_mlp_factored(bu.pre, bu.ctr, bu.nbr2, bu.cnt2, B, NP1, NP2, NS, w->sa2_pack, C1, 128, 128, C2,
                              bu.sa3_in + 3, K3, stream));
  // ---- group-all module: B >= SA3_CHAIN_MIN_BATCH problems -> the fused chain (nothing between the rows and the pooled
  // row touches HBM); fewer: three dense layers over the B*128 rows, max over each environment's rows -------------------
  if (B >= SA3_CHAIN_MIN_BATCH && w->sa3_pack) {  // (no pack: the layer-by-layer form below)
    MPX_TRY(mpx_sa3_chain(bu.sa3_in, K3, B, NP2, w->sa3_pack, K3, H3, H3, C3, bu.pooled, C3, stream));
  } else {
  MPX_TRY(lin(bu.sa3_in, K3, w->sa3_w[0], w->sa3_b[0], B * NP2, H3, K3, MPX_ACT_RELU, bu.h_a, H3));
  MPX_TRY(lin(bu.h_a, H3, w->sa3_w[1], w->sa3_b[1], B * NP2, H3, H3, MPX_ACT_RELU, bu.h_b, H3));
  if (B > 8) {
    MPX_TRY(mpx_linear_rowmax(bu.h_b, H3, w->sa3_w[2], w->sa3_b[2], B * NP2, C3, H3, NP2, bu.pooled, C3, stream));
  } else {  // a handful of problems: split-K layer + row-max (see model.py)
    MPX_TRY(lin(bu.h_b, H3, w->sa3_w[2], w->sa3_b[2], B * NP2, C3, H3, MPX_ACT_RELU, bu.h_a, C3));
    MPX_TRY(mpx_rowmax(bu.h_a, C3, B, NP2, C3, bu.pooled, C3, stream));
  }
  }
  // ---- fc head: 1024 -> 4096 -> GN -> 2048 -> GN -> 2048 (into the left part of the decoder's input rows) ------
  MPX_TRY(lin(bu.pooled, C3, w->fc_w[0], w->fc_b[0], B, 4096, C3, MPX_ACT_NONE, bu.fc_a, 4096));
  MPX_TRY(mpx_groupnorm_leaky(bu.fc_a, w->gn_g[0], w->gn_b[0], B, 4096, 16, 1e-5f, bu.fc_a, stream));
  MPX_TRY(lin(bu.fc_a, 4096, w->fc_w[1], w->fc_b[1], B, 2048, 4096, MPX_ACT_NONE, bu.fc_b, 2048));
  MPX_TRY(mpx_groupnorm_leaky(bu.fc_b, w->gn_g[1], w->gn_b[1], B, 2048, 16, 1e-5f, bu.fc_b, stream));
  MPX_TRY(lin(bu.fc_b, 2048, w->fc_w[2], w->fc_b[2], B, ENC, 2048, MPX_ACT_NONE, bu.cat, CAT));
  // ---- decoder 2112 -> 512 -> 256 -> 128 -> 7 ------------------------------------------------------------------
  MPX_TRY(lin(bu.cat, CAT, w->de_w[0], w->de_b[0], B, 512, CAT, MPX_ACT_LEAKY, bu.s_a, 512));
  MPX_TRY(lin(bu.s_a, 512, w->de_w[1], w->de_b[1], B, 256, 512, MPX_ACT_LEAKY, bu.s_b, 256));
  MPX_TRY(lin(bu.s_b, 256, w->de_w[2], w->de_b[2], B, 128, 256, MPX_ACT_LEAKY, bu.s_a, 128));
  MPX_TRY(lin(bu.s_a, 128, w->de_w[3], w->de_b[3], B, 7, 128, MPX_ACT_NONE, dq, 7));
  MPX_LAUNCH_CHECK("mpx_policy_forward");
}

// ---- closed-loop rollouts: RolloutEngine.step() / .rollout() / the body of rollout_until_success -----------------
namespace {
int64_t pad256(int64_t n) { return (n + 255) / 256 * 256; }
}  // namespace

MPX_EXPORT int64_t mpx_rollout_workspace(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  // policy workspace | dq [B,7] | obstacle ids of a scene re-render [B, N] uint16 (at most N scene rows)
  return mpx_policy_workspace(B, N) + pad256((int64_t)B * 7 * (int64_t)sizeof(float)) + pad256((int64_t)B * N * 2);
}

MPX_EXPORT int mpx_rollout(const mpx_policy_weights *w, const mpx_rollout_scene *sc, const mpx_rollout_options *opt,
                           float *xyz, int N, float *q_norm, float *q, int B, int32_t *flags, float *min_sdf,
                           void *workspace, int64_t workspace_bytes, mpx_stream_t stream) {
  MPX_REQUIRE(w && sc && opt && xyz && q_norm && q && flags, "mpx_rollout: NULL operand");
  MPX_REQUIRE(sc->n_robot >= 1 && sc->n_robot <= N, "mpx_rollout: n_robot = %d outside [1, N]", sc->n_robot);
  MPX_REQUIRE(opt->steps >= 0 && opt->first_step >= 0, "mpx_rollout: negative step count");
  const bool rerender = opt->n_scene > 0;
  if (rerender) {
    MPX_REQUIRE(sc->n_robot + opt->n_scene <= N, "mpx_rollout: %d robot + %d scene rows exceed the slab's %d", sc->n_robot,
                opt->n_scene, N);
    MPX_REQUIRE(opt->cub_centers && opt->cub_quats && opt->cyl_centers && opt->cyl_quats,
                "mpx_rollout: a scene re-render needs the primitives' centres and quaternions");
  }
  MPX_REQUIRE(!opt->target_poses || opt->done, "mpx_rollout: success tracking needs the done flags");
  MPX_REQUIRE(!opt->trajectory || (opt->trajectory_row >= 0 && opt->trajectory_len >= opt->trajectory_row + opt->steps),
              "mpx_rollout: trajectory rows hold %d waypoints, rows [%d, %d) are written", opt->trajectory_len,
              opt->trajectory_row, opt->trajectory_row + opt->steps);
  MPX_REQUIRE(opt->subset_table_size == 0 || (opt->subset_buf && opt->subset_table_size >= sc->n_robot),
              "mpx_rollout: per-step subset needs subset_buf and a table of >= n_robot rows");
  if (B == 0 || opt->steps == 0) return 0;
  const int64_t need = mpx_rollout_workspace(B, N);
  MPX_REQUIRE(workspace && workspace_bytes >= need, "mpx_rollout: workspace of %lld bytes, mpx_rollout_workspace asks for %lld",
              (long long)workspace_bytes, (long long)need);
  const int64_t policy_bytes = mpx_policy_workspace(B, N);
  char *base = static_cast<char *>(workspace);
  float *dq = reinterpret_cast<float *>(base + policy_bytes);
  uint16_t *assign = reinterpret_cast<uint16_t *>(base + policy_bytes + pad256((int64_t)B * 7 * (int64_t)sizeof(float)));
  for (int i = 0; i < opt->steps; ++i) {
    const int step = opt->first_step + i;
    if (rerender)  // a fresh scene cloud from the primitives (seed schedule of RolloutEngine: scene_seed + 7919 * step)
      MPX_TRY(mpx_scene_cloud(opt->cub_centers, sc->cub_dims, opt->cub_quats, sc->M1, opt->cyl_centers, sc->cyl_radii,
                              sc->cyl_heights, opt->cyl_quats, sc->M2, B, opt->n_scene,
                              opt->scene_seed + 7919ull * (uint64_t)step, opt->env_offset, assign, nullptr, nullptr,
                              xyz + (int64_t)sc->n_robot * 4, (int64_t)N * 4, 4, 0, stream));
    MPX_TRY(mpx_policy_forward(w, xyz, N, q_norm, B, dq, workspace, policy_bytes, stream));
    MPX_TRY(mpx_joint_step(q_norm, dq, sc->limits, B, q_norm, q, opt->target_poses ? opt->done : nullptr, stream));
    if (opt->target_poses)  // 1 cm / 15 deg early-stop test (run_inference.py:176-187): finished environments freeze
      MPX_TRY(mpx_franka_success(q, opt->target_poses, B, sc->finger, opt->pos_tol, opt->cos_rot_tol, opt->done,
                                 opt->steps_taken, nullptr, nullptr, stream));
    const int32_t *subset = sc->subset;
    if (opt->subset_table_size > 0) {  // the reference redraws the robot-point subset on every sampler call
      MPX_TRY(mpx_draw_subset(opt->subset_table_size, sc->n_robot, opt->subset_seed, step, opt->subset_buf, stream));
      subset = opt->subset_buf;
    }
    MPX_TRY(mpx_franka_cloud(q, B, sc->finger, sc->table_pts, sc->table_link, subset, sc->n_robot, xyz, (int64_t)N * 4, 4,
                             stream));
    MPX_TRY(mpx_franka_collision(q, B, 1, sc->finger, sc->sph_centers, sc->sph_radii, sc->sph_link, sc->n_spheres,
                                 sc->cub_frames, sc->cub_dims, sc->M1, sc->cyl_frames, sc->cyl_radii, sc->cyl_heights, sc->M2,
                                 flags, min_sdf, stream));
    if (opt->trajectory) {  // waypoint trajectory_row + i of every environment
      hipError_t e = hipMemcpy2DAsync(opt->trajectory + (int64_t)(opt->trajectory_row + i) * 7, (size_t)opt->trajectory_len * 7 * sizeof(float),
                                      q, 7 * sizeof(float), 7 * sizeof(float), (size_t)B, hipMemcpyDeviceToDevice,
                                      mpx_s(stream));
      MPX_REQUIRE(e == hipSuccess, "mpx_rollout: trajectory copy failed: %s", hipGetErrorString(e));
    }
  }
  return 0;
}

MPX_EXPORT int mpx_rollout_step(const mpx_policy_weights *w, const mpx_rollout_scene *sc, float *xyz, int N, float *q_norm,
                                float *q, int B, int32_t *flags, float *min_sdf, void *workspace, int64_t workspace_bytes,
                                mpx_stream_t stream) {
  mpx_rollout_options opt = {};
  opt.steps = 1;
  return mpx_rollout(w, sc, &opt, xyz, N, q_norm, q, B, flags, min_sdf, workspace, workspace_bytes, stream);
}
