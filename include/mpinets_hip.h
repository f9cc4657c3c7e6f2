/*
 * mpinets_hip.h -- C-ABI of libmpinets_hip.so, the MI355X (gfx950) engine behind the hot path
 * of NVlabs/motion-policy-networks.
 *
 * The reference has no FFI of its own (it is pure Python); the seams this ABI
 * sits behind are the Python-level call sites cited on each entry point
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; plain C types only;
 *   - all tensors are dense, row-major, float32 / int32; "stride" arguments count floats;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue (the first launch of
 *     a large-LDS kernel on a device also raises that kernel's LDS limit, once);
 *   - return 0 on success, non-zero on error (mpx_last_error() gives the text, per thread);
 *   - no call allocates or frees device memory, so every call is hipGraph-capturable
 *     (exception: none).
 */
#ifndef MPINETS_HIP_H
#define MPINETS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mpx_stream_t;

#define MPX_NUM_FRAMES 15 /* link0..8, hand, leftfinger, rightfinger, l/r fingertip, right_gripper */

int mpx_version(void); /* 340: mpx_pool_wgrad / mpx_pool_wgrad_scratch / mpx_pool_dgrad, mpx_linear_segmax / mpx_linear_segmax_bf16x3, mpx_pack_rows_ld / mpx_pack_rows_grad_ld (additions only);
                          330: mpx_sa3_front_bf16x3 / _pack / _pack_size / _w3_pairs (additions), the measurement hooks mpx_sa3_chain_probe / mpx_sa2_bf16x3_set_probe /
                          mpx_sa3_front_bf16x3_probe declared, mpx_sa_mlp_bf16x3_factored refuses nsample > 128;
                          320: mpx_linear_dact, mpx_segment_max_grad_act, mpx_linear_bf16x3_dact, mpx_linear_wgrad_bf16x3 (additions only); mpx_franka_collision accepts
                          frame pointers that are not 16-byte aligned;
                          310: struct mpx_policy_weights ends with sa3_pack (NULL = layer-by-layer group-all module at every
                          batch size; a caller built against the 200 header must be rebuilt), MPX_VARIANT_UNIT_QUEUE,
                          mpx_ball_query_hits rejects nsample > 256;
                          300: mpx_set_variant / mpx_get_variant, mpx_sa_mlp_bf16x3_wants_order takes nsample */
const char *mpx_last_error(void);
/* host-side query of the device the library will launch on (name buffer may be NULL) */
int mpx_device_info(char *name, int name_len, int *cu_count, int *lds_bytes);

/* ---- geometry: mpinets/geometry.py -------------------------------------------------------- */

/* TorchCuboids/TorchCylinders.__init__ + _init_frames (geometry.py:151,177-223,409-454).
 * centers [n,3], quats [n,4] (w,x,y,z; normalised here) -> inv_frames [n,4,4] row-major:
 * rows 0..2 = [R | R.(-c)], row 3 = [0 0 0 1].  R is the reference's matrix as written,
 * including `yz - wx` at both R[1][2] and R[2][1].                                          */
int mpx_prim_frames(const float *centers, const float *quats, int n, float *inv_frames,
                    mpx_stream_t stream);

/* TorchCuboids.sdf / .sdf_sequence (geometry.py:238-288, :290-347).
 * points [B,P,3] (P = N, or T*N), out [B,P] = min over unmasked cuboids; cuboids with any
 * |dim| <= 1e-8 are masked; all masked -> +inf.                                             */
int mpx_cuboid_sdf(const float *inv_frames, const float *dims, int B, int M, const float *points,
                   int P, float *out, mpx_stream_t stream);
/* TorchCylinders.sdf / .sdf_sequence (geometry.py:456-507, :509-568). radii/heights [B,M]. */
int mpx_cylinder_sdf(const float *inv_frames, const float *radii, const float *heights, int B,
                     int M, const float *points, int P, float *out, mpx_stream_t stream);
/* TorchSpheres.sdf / .sdf_sequence (geometry.py:87-123). centers [B,M,3], radii [B,M].     */
int mpx_sphere_sdf(const float *centers, const float *radii, int B, int M, const float *points,
                   int P, float *out, mpx_stream_t stream);

/* ---- robot geometry: robofin FrankaSampler / FrankaCollisionSampler call sites ------------ */

/* FK of the Franka Panda chain: q [B,7] -> frames [B,15,12] (R row-major 3x3, then t).
 * Replaces FrankaSampler.end_effector_pose / FrankaRobot.fk (model.py:275,
 * run_inference.py:176-178).  `finger` = prismatic finger opening (0.025 in the reference). */
int mpx_franka_fk(const float *q, int B, float finger, float *frames, mpx_stream_t stream);

/* FrankaSampler.sample (model.py:250, run_inference.py:64,111,169, data_loader.py:180-185):
 * out[b, j, 0:3] = frame[b, link[src]] * table[src], src = subset ? subset[j] : j.
 * out is addressed as out + b*out_batch_stride + j*out_point_stride (floats), which lets the
 * call write straight into the xyz slab `xyz[:, :n_out, :3]` (model.py:180-181).            */
int mpx_franka_cloud(const float *q, int B, float finger, const float *table_pts,
                     const int32_t *table_link, const int32_t *subset, int n_out, float *out,
                     int64_t out_batch_stride, int out_point_stride, mpx_stream_t stream);

/* FrankaSampler.sample_end_effector (run_inference.py:66-69, data_loader.py:158-161):
 * poses [B,4,4] row-major; out[b,j] = pose[b] * table[src].                                 */
int mpx_pose_cloud(const float *poses, int B, const float *table_pts, const int32_t *subset,
                   int n_out, float *out, int64_t out_batch_stride, int out_point_stride,
                   mpx_stream_t stream);

/* FrankaCollisionSampler.compute_spheres (model.py:300): centres [B,S,3] of the S table
 * spheres (sph_centers [S,3] link-local, sph_link [S]) at configuration q [B,7].           */
int mpx_franka_spheres(const float *q, int B, float finger, const float *sph_centers,
                       const int32_t *sph_link, int S, float *out, mpx_stream_t stream);

/* Fused swept-sphere collision check, model.py:293-314, for q [B,T,7]:
 *   flags[b] |= any_{t,s} min(cuboid_sdf, cylinder_sdf)(centre[b,t,s]) <= sph_radii[s]
 * flags int32 [B] is OR-ed into (caller zeroes it); min_sdf [B,T,S] optional (may be NULL).
 * Either primitive set may be empty (M = 0).  The frame arrays are read fastest when they are
 * 16-byte aligned (the classes' inv_frames always are); unaligned pointers are accepted and
 * take a slower kernel with the same results.                                                */
int mpx_franka_collision(const float *q, int B, int T, float finger, const float *sph_centers,
                         const float *sph_radii, const int32_t *sph_link, int S,
                         const float *cub_frames, const float *cub_dims, int M1,
                         const float *cyl_frames, const float *cyl_radii,
                         const float *cyl_heights, int M2, int32_t *flags, float *min_sdf,
                         mpx_stream_t stream);

/* rollout joint update, model.py:171-173 + utils.py:207-209:
 *   q_norm_out = clamp(q_norm + dq, -1, 1);  q_out = (q_norm_out + 1) * (hi - lo) / 2 + lo
 * limits [7,2] device; either output may alias q_norm; environments with frozen[b] != 0
 * (optional int32 [B]) keep their configuration.                                            */
int mpx_joint_step(const float *q_norm, const float *dq, const float *limits, int B,
                   float *q_norm_out, float *q_out, const int32_t *frozen, mpx_stream_t stream);

/* early-stop test of rollout_until_success (run_inference.py:176-187), on the device:
 *   success = |t(right_gripper(q)) - t(target)| < pos_tol  &&  cos(angle(R R_t^T)) > cos_rot_tol
 * target_poses [B,4,4] row-major.  done int32 [B] is OR-ed (a set flag freezes that environment in
 * mpx_joint_step); steps (optional) counts the policy steps taken until done; pos_err / cos_angle
 * (optional) [B] return the two measured quantities.                                            */
int mpx_franka_success(const float *q, const float *target_poses, int B, float finger, float pos_tol,
                       float cos_rot_tol, int32_t *done, int32_t *steps, float *pos_err,
                       float *cos_angle, mpx_stream_t stream);

/* Batched trajectory metrics (row N3; mpinets/metrics.py:311-384, 410-434 without PyBullet):
 * traj [B,T,7] joint angles, lengths (optional int32 [B]) valid waypoints per trajectory,
 * target_poses [B,4,4] (right_gripper), limits [7,2].  Per trajectory: final position error [cm],
 * final orientation error [deg], end-effector path lengths (m, deg), joint-limit violation flag,
 * self-collision flag.  Self collision uses the in-repo Geometric-Fabrics model
 * (config/franka_fabric_config.yaml:120-140: base body cylinder vs spheres on link7 / hand /
 * finger tips) -- NOT PyBullet's mesh test like the reference Evaluator.                        */
int mpx_trajectory_metrics(const float *traj, const int32_t *lengths, const float *target_poses,
                           const float *limits, int B, int T, float finger, float *pos_err_cm,
                           float *orient_err_deg, float *path_pos, float *path_orient_deg,
                           int32_t *limit_violation, int32_t *self_collision, mpx_stream_t stream);

/* ---- split-bf16 ("bf16x3") dense layers: the opt-in fast mode of mpx_linear / mpx_linear_rowmax ------
 * Every fp32 product is evaluated as x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on the bf16 matrix cores (fp32
 * accumulate).  Split operands are held in the PAIRS form: a row holds, per group of 16 k-values,
 * [hi x 16 | lo x 16] bf16 (hi = bf16(v), lo = bf16(v - hi)), K padded with zeros to Kp = roundup(K, 16):
 * [rows, ld >= 2 Kp] bf16 -- the same bytes as the fp32 rows.  mpx_split_bf16 converts fp32 rows [R, K]
 * (weights [N, K], once; or activations) into pairs.  Same argument meaning as the fp32 entry points otherwise;
 * weights are always passed as pairs with ld = 2 Kp.
 *   mpx_linear_bf16x3 / mpx_linear_rowmax_bf16x3: fp32 activations (split while staged), fp32 result.
 *   mpx_linear_bf16x3_to_pairs: the same, result written as pairs (N and ldp multiples of 4, ldp >= 2 Np).
 *   mpx_linear_bf16x3_pairs: activations already in pairs (a_pairs [M, lda], K a multiple of 16, lda of 8; under
 *     4 GB), result EITHER fp32 rows (y) OR pairs (y_pairs); the other pointer NULL.
 *   mpx_linear_rowmax_bf16x3_pairs: ReLU + max over each group of rows = 128 rows, as fp32 (y) or pairs (y_pairs).
 * Chains of layers (the group-all module's 259 -> 512 -> 512 -> 1024 MLP, model.py:383) keep their activations in
 * pairs: a value is split once, by the epilogue that produces it, and every form accumulates in the same order, so a
 * chain through pairs equals the chain through fp32 rows bit for bit.                                            */
int mpx_split_bf16(const float *x, int ldx, int64_t R, int K, void *pairs, int ldp, mpx_stream_t stream);
int mpx_linear_bf16x3(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N, int K,
                      int act, float *y, int ldy, mpx_stream_t stream);
int mpx_linear_rowmax_bf16x3(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N,
                             int K, int rows, float *y, int ldy, mpx_stream_t stream);
int mpx_linear_bf16x3_to_pairs(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N,
                               int K, int act, void *y_pairs, int ldp, mpx_stream_t stream);
int mpx_linear_bf16x3_pairs(const void *a_pairs, int lda, const void *w_pairs, const float *bias, int M, int N,
                            int K, int act, float *y, int ldy, void *y_pairs, int ldp, mpx_stream_t stream);
int mpx_linear_rowmax_bf16x3_pairs(const void *a_pairs, int lda, const void *w_pairs, const float *bias, int M,
                                   int N, int K, int rows, float *y, int ldy, void *y_pairs, int ldp,
                                   mpx_stream_t stream);

/* The group-all module's first two layers (272 -> 512 -> 512, ReLU each; model.py:377-383) in bf16x3 as ONE kernel: a
 * workgroup per environment, the rows divided among its waves, activations in registers from layer to layer, the weights
 * streamed through an LDS ring once per environment (csrc/sa3_front_bf16.hip).  x: fp32 rows [B * 128, ldx >= 272]
 * ([xyz | f | 0], 16-byte aligned); y_pairs: [B * 128, ldp >= 1024] bf16 in the pairs form with the k-steps in the KERNEL's
 * channel order -- the operand of mpx_linear_rowmax_bf16x3_pairs with the weight pairs of mpx_sa3_front_bf16x3_w3_pairs
 * (the same columns permuted).  Equal to mpx_linear_bf16x3_to_pairs + mpx_linear_bf16x3_pairs to rounding (the 16 products
 * of a k-step enter the MFMA in another order, the bias is added first), not bit for bit.
 * pack: mpx_sa3_front_bf16x3_pack_size bytes (host call; -1: unsupported widths), written by mpx_sa3_front_bf16x3_pack from
 * the fp32 weights w1 [c1, k1_real <= K1], w2 [c2, c1] and biases.                                                    */
int64_t mpx_sa3_front_bf16x3_pack_size(int K1, int c1, int c2);
int mpx_sa3_front_bf16x3_pack(const float *w1, int k1_real, const float *b1, const float *w2, const float *b2, int K1,
                              int c1, int c2, void *pack, mpx_stream_t stream);
int mpx_sa3_front_bf16x3_w3_pairs(const float *w3, int c3, int c2, void *w3_pairs, mpx_stream_t stream);
int mpx_sa3_front_bf16x3(const float *x, int ldx, int B, int rows, const void *pack, void *y_pairs, int ldp,
                         mpx_stream_t stream);

/* ---- training losses with analytic gradients (row N1; mpinets/loss.py:31-166) -------------------- */

/* collision_loss (loss.py:48-95) on points [B,N,3] (strides in floats): per environment
 * loss_sum[b] = sum_n max(0, margin - min(cuboid sdf, cylinder sdf)); the reference's mean is
 * sum_b loss_sum[b] / (B*N).  grad_points (optional, strided like points) receives
 * d(loss_sum[b]) / d(point): the autograd of geometry.py:256-288 / :478-507 restated analytically.
 * frames / dims / radii / heights as for mpx_cuboid_sdf / mpx_cylinder_sdf.                          */
int mpx_collision_hinge(const float *points, int64_t batch_stride, int point_stride, int B, int N,
                        const float *cub_frames, const float *cub_dims, int M1, const float *cyl_frames,
                        const float *cyl_radii, const float *cyl_heights, int M2, float margin,
                        float *loss_sum, float *grad_points, int64_t grad_batch_stride,
                        int grad_point_stride, mpx_stream_t stream);

/* point_match_loss (loss.py:31-45) on contiguous input/target [B, n_per_env]: sums[b] =
 * {sum d^2, sum |d|}; grad_input (optional) = w_sq*2*d + w_abs*sign(d) -- pass w = 1/(B*n_per_env)
 * for the reference's two mean reductions.                                                           */
int mpx_point_match(const float *input, const float *target, int B, int n_per_env, float w_sq, float w_abs,
                    float *sums, float *grad_input, mpx_stream_t stream);

/* Backward of mpx_franka_cloud (robofin FrankaSampler.sample under autograd; loss.py:142-147):
 * grad_q[b,k] = sum_points grad_points[b,p] . d(point)/d(q_k), joint angles in radians.             */
int mpx_franka_cloud_grad(const float *q, int B, float finger, const float *table_pts,
                          const int32_t *table_link, const int32_t *subset, int n, const float *grad_points,
                          int64_t grad_batch_stride, int grad_point_stride, float *grad_q,
                          mpx_stream_t stream);

/* ---- backward of the dense layers (row N1): y = act(x . W^T + b) ------------------------------------
 * dX is the forward kernel on the transposed weights: mpx_linear(dZ, W^T).  mpx_act_backward forms
 * dZ = dY * act'(y) from the layer OUTPUT y (contiguous, n elements).  mpx_linear_wgrad computes
 * dW [N,K] = dZ^T . x and db [N] = column sums of dZ (optional) with the reduction over the M rows split across
 * workgroups; the splits' partial tiles go through `scratch` (mpx_linear_wgrad_scratch(M,N,K) floats) and are
 * added in a fixed order (deterministic).  N, K, ldx, lddy multiples of 4.                                  */
int mpx_act_backward(const float *dy, const float *y, int64_t n, int act, float *dz, mpx_stream_t stream);
/* dX with the previous layer's elementwise backward in the GEMM epilogue: y [M,N] = (x [M,K] . w [N,K]^T) * act'(dact_of)
 * (dact_of [M,N] = that layer's OUTPUT rows, leading dimension lddact; dact = its activation; MPX_ACT_NONE: plain
 * product).  Same arithmetic as mpx_linear(x, w) followed by mpx_act_backward.  loss.py / model.py:185-240 via autograd. */
int mpx_linear_dact(const float *x, int ldx, const float *w, int M, int N, int K, const float *dact_of,
                    int lddact, int dact, float *y, int ldy, mpx_stream_t stream);
/* The training GEMMs in the split-bf16 arithmetic of the `bf16x3` mode (three bf16 MFMAs per fp32 product, fp32
 * accumulate, fp32 master weights: the engine's form of the reference's precision=16, run_training.py:112): the forward is
 * mpx_linear_bf16x3; dX = mpx_linear_bf16x3_dact (w_pairs = mpx_split_bf16 of the TRANSPOSED weights); dW / db =
 * mpx_linear_wgrad_bf16x3 (same arguments, scratch and split reduction as mpx_linear_wgrad).                           */
int mpx_linear_bf16x3_dact(const float *x, int ldx, const void *w_pairs, int M, int N, int K, const float *dact_of,
                           int lddact, int dact, float *y, int ldy, mpx_stream_t stream);
int mpx_linear_wgrad_bf16x3(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, float *dw,
                            float *db, float *scratch, mpx_stream_t stream);
int64_t mpx_linear_wgrad_scratch(int M, int N, int K);
int mpx_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, int M, int N, int K, float *dw,
                     float *db, float *scratch, mpx_stream_t stream);

/* backward of mpx_groupnorm_leaky on [M, C]: dx, dgamma [C], dbeta [C]; stats = scratch of 2*M*groups floats    */
int mpx_groupnorm_leaky_grad(const float *x, const float *gamma, const float *beta, const float *dy, int M,
                             int C, int groups, float eps, float *dx, float *dgamma, float *dbeta,
                             float *stats, mpx_stream_t stream);

/* ---- differentiable grouping + max-pool of the set-abstraction stack (row N1) --------------------
 * The reference trains through pointnet2_ops' QueryAndGroup / max-pool (model.py:366-383).  Here a
 * neighbourhood contributes only its distinct neighbours (cnt from mpx_ball_query; padding repeats
 * the first hit and neither a max nor its gradient sees repeats): rows of one [R, 3+C] matrix,
 * query q owning rows offsets[q] .. offsets[q+1] (offsets int64 [B*npoint+1] = exclusive prefix
 * sum of max(cnt,1)).                                                                              */

/* rows[offsets[q]+r] = [xyz[idx[q,r]] - new_xyz[q] | feat[idx[q,r]]]  (QueryAndGroup, use_xyz=True) */
int mpx_pack_rows(const float *xyz, int xyz_stride, const float *new_xyz, int new_stride,
                  const float *feat, int feat_stride, int C, const int32_t *idx, const int32_t *cnt,
                  const int64_t *offsets, int B, int N, int npoint, int nsample, float *rows,
                  mpx_stream_t stream);
/* ... with the rows `ld` >= 3 + C floats apart and the columns behind 3 + C zero-filled (ld % 4 == 0: no padding copy in
 * front of the GEMMs)                                                                                 */
int mpx_pack_rows_ld(const float *xyz, int xyz_stride, const float *new_xyz, int new_stride, const float *feat,
                     int feat_stride, int C, const int32_t *idx, const int32_t *cnt, const int64_t *offsets, int B, int N,
                     int npoint, int nsample, float *rows, int ld, mpx_stream_t stream);
/* backward: grad_feat[b, idx[q,r], c] += grad_rows[offsets[q]+r, 3+c] (atomic adds; grad_feat pre-zeroed) */
int mpx_pack_rows_grad(const float *grad_rows, int C, const int32_t *idx, const int32_t *cnt,
                       const int64_t *offsets, int B, int N, int npoint, int nsample, float *grad_feat,
                       int feat_stride, mpx_stream_t stream);
int mpx_pack_rows_grad_ld(const float *grad_rows, int ld, int C, const int32_t *idx, const int32_t *cnt,
                          const int64_t *offsets, int B, int N, int npoint, int nsample, float *grad_feat, int feat_stride,
                          mpx_stream_t stream); /* (gradient rows ld >= 3 + C floats apart) */
/* out[q,c] = max over the rows of segment q of y[R,C]; arg[q,c] = first row attaining it           */
int mpx_segment_max(const float *y, int C, const int64_t *offsets, int64_t Q, float *out, int out_stride,
                    int64_t *arg, mpx_stream_t stream);
/* backward: grad_y[arg[q,c], c] = grad_out[q,c] (grad_y pre-zeroed)                                  */
int mpx_segment_max_grad(const float *grad_out, int grad_stride, const int64_t *arg, int64_t Q, int C,
                         float *grad_y, mpx_stream_t stream);
/* the same with the backward of the activation in front of the pool folded in: out [Q, >= C] = the pooled rows
 * (= the activation's output at the arg-max): grad_y[r, c] = (r == arg[q,c]) ? grad_out[q,c] * act'(out[q,c]) : 0 for
 * EVERY row r of segment q -- grad_y needs no zero fill                                                         */
int mpx_segment_max_grad_act(const float *grad_out, int grad_stride, const int64_t *arg, const float *out,
                             int out_stride, const int64_t *offsets, int64_t Q, int C, int act, float *grad_y,
                             mpx_stream_t stream);
/* Forward of [dense layer + activation + segment max-pool] without the [M, N] matrix in memory: pooled[q, n] = max over the
 * rows r with seg[r] == q of act(x[r] . w[n] + bias[n]), arg[q, n] = the first such row -- mpx_linear followed by
 * mpx_segment_max, bit for bit (the same tile kernel; its epilogue folds the tile into 64-bit {value, ~row} keys with
 * atomicMax).  seg int32 [M]: the segment of every row, non-decreasing, every q in [0, Q) present; keys: Q*N*8 bytes of
 * scratch (8-byte aligned); pooled [Q, >= N] (ldp floats apart), arg int64 [Q, N].  _bf16x3: the product in split bf16
 * (w_pairs from mpx_split_bf16, as mpx_linear_bf16x3).  (Reference: pointnet2_ops' SharedMLP + max_pool2d, model.py:366-383.) */
int mpx_linear_segmax(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K, int act,
                      const int32_t *seg, int64_t Q, void *keys, float *pooled, int ldp, int64_t *arg, mpx_stream_t stream);
int mpx_linear_segmax_bf16x3(const float *x, int ldx, const void *w_pairs, const float *bias, int M, int N, int K, int act,
                             const int32_t *seg, int64_t Q, void *keys, float *pooled, int ldp, int64_t *arg,
                             mpx_stream_t stream);
/* Backward of [dense layer (W [C,K], b) + activation `act` + segment max-pool] that never forms the [R, C] gradient in
 * front of the pool: per (query, channel) only the arg-max row carries gz[q,c] = grad_out[q,c] * act'(out[q,c]) (out = the
 * pooled rows).  x [R, >= K] = the layer's INPUT rows (ldx floats apart), arg = mpx_segment_max's (global row numbers).
 *   mpx_pool_wgrad: dw[c,k] = sum_q gz[q,c] * x[arg[q,c], k], db[c] = sum_q gz[q,c]  (db NULL or == dw + C*K: one fixed-order
 *                   reduction over the query splits; scratch = mpx_pool_wgrad_scratch(Q, C, K) floats);
 *   mpx_pool_dgrad: gx[r,k] = below'(x[r,k]) * sum_{c: arg[q,c] == r} gz[q,c] * w[c,k] for EVERY row r of every segment (no
 *                   zero fill needed; `below` = activation code of the layer that produced x, 0 = none; max_rows = the
 *                   longest segment the caller expects (a hint, any value works); C <= 65535).
 * fp32 FMAs in a fixed order: deterministic; the same sums as mpx_segment_max_grad_act + mpx_linear_wgrad / mpx_linear_dact
 * in another summation order.  (Reference: the autograd of pointnet2_ops' QueryAndGroup + max_pool2d, model.py:366-383.) */
int64_t mpx_pool_wgrad_scratch(int64_t Q, int C, int K);
int mpx_pool_wgrad(const float *grad_out, int grad_stride, const int64_t *arg, const float *out, int out_stride,
                   int64_t Q, int C, int act, const float *x, int ldx, int K, float *dw, float *db, float *scratch,
                   mpx_stream_t stream);
int mpx_pool_dgrad(const float *grad_out, int grad_stride, const int64_t *arg, const float *out, int out_stride,
                   const int64_t *offsets, int64_t Q, int C, int act, const float *w, int ldw, const float *x, int ldx,
                   int below, int K, int max_rows, float *gx, int ldg, mpx_stream_t stream);

/* ---- batch assembly from the dataset arrays (row N2; mpinets/data_loader.py:141-280, 390-417) ------- */

/* Per-sample joint quantities of PointCloudBase.get_inputs for a whole batch.  trajectories
 * [n_traj, L, 7] (the HDF5 `hybrid_solutions` / `global_solutions` array, resident in HBM), traj_idx
 * int64 [B] (clamped to [0, n_traj)), timestep int32 [B] (NULL = 0: the validation dataset).  Outputs: q [B,7] =
 * the waypoint (+ N(0, noise_scale) joint noise when noise_scale > 0; Philox keyed by (seed, sample_offset + row),
 * so a shard of a batch draws what the whole batch draws), clamped to the limits whenever `train` != 0
 * (data_loader.py:176-178 clamps every TRAIN sample, with or without noise), q_norm = normalize(q), sup_norm (optional) = normalize(waypoint min(t+1, L-1)),
 * target_pose [B,4,4] / target_pos [B,3] = right_gripper FK of the LAST waypoint.                  */
int mpx_batch_configs(const float *trajectories, int64_t n_traj, int L, const int64_t *traj_idx,
                      const int32_t *timestep, const float *limits, float noise_scale, uint64_t seed,
                      int64_t sample_offset, int train, int B, float finger, float *q, float *q_norm, float *sup_norm, float *target_pose,
                      float *target_pos, mpx_stream_t stream);
/* dst[b, :] = src[idx[b], :] for rows of row_floats floats (primitive rows of the sampled scenes)  */
int mpx_gather_rows(const float *src, const int64_t *idx, int B, int row_floats, float *dst,
                    mpx_stream_t stream);

/* ---- partial-view scene clouds from a depth camera (row N4; run_inference.py:194-257) ------------- */

/* Analytic ray casting of the primitives from cam_poses [B,4,4] (world-from-camera, OpenGL axes: x right,
 * y up, looking along -z, as the reference's evaluation poses are given), pinhole intrinsics in pixels,
 * image W x H.  depth [B, H*W] = distance along the pixel's ray to the nearest cuboid / cylinder, or -1 when
 * nothing is hit within far_clip or one of the robot's spheres (sph_centers [B,S,3] from mpx_franka_spheres,
 * sph_radii [S]; S = 0: no robot) is in front -- the reference removes the robot's pixels.
 * Replaces PyBullet's rasteriser + depth buffer (robofin Bullet.get_pointcloud_from_camera).       */
int mpx_depth_render(const float *cam_poses, float fx, float fy, float cx, float cy, int W, int H, int B,
                     const float *cub_frames, const float *cub_dims, int M1, const float *cyl_frames,
                     const float *cyl_radii, const float *cyl_heights, int M2, const float *sph_centers,
                     const float *sph_radii, int S, float far_clip, float *depth, mpx_stream_t stream);
/* np.random.choice(len(cloud), n_out, replace=False) of run_inference.py:78-85 on the device: every valid
 * pixel gets a Philox4x32-10 key (seed, env_offset + environment, pixel); the n_out smallest keys are written as world
 * points in key order to out (strides in floats).  count [B] = valid pixels; an environment with fewer than
 * n_out of them is left untouched (the caller raises like numpy).  n_out <= 4096.                    */
int mpx_depth_select(const float *depth, const float *cam_poses, float fx, float fy, float cx, float cy,
                     int W, int H, int B, int n_out, uint64_t seed, int64_t env_offset, float *out,
                     int64_t out_batch_stride, int out_point_stride, int32_t *count, mpx_stream_t stream);

/* ---- scene point clouds: mpinets/geometry.py:571-608 (construct_mixed_point_cloud), batched ----- */

/* For every environment: area-proportional pool sizes int(p_i*N)+500, N pool slots drawn without
 * replacement in random order (every slot gets a Philox key, the N smallest keys win in key order;
 * only the owning obstacle matters), one fresh uniform surface sample per slot, labels = shuffled
 * 1..K.  N <= 4096.  Counter RNG Philox4x32-10 keyed by
 * (seed, GLOBAL environment id = env_offset + row): results depend on nothing else, so rank r of a sharded batch
 * (env_offset = its first global environment) draws exactly what one process would for those environments.  Zero-volume primitives are skipped;
 * obstacle ids count cuboids first, then cylinders.  An environment without obstacles gets
 * ids 0xFFFF and zero points (the reference returns an empty array, geometry.py:586-587).
 *   assign uint16 [B,N] (required scratch/output), labels uint8 [B,M1+M2] (optional),
 *   n_obstacles int32 [B] (optional); out rows at out + b*out_batch_stride + j*out_point_stride
 *   receive x,y,z (and the label as float in column 3 when write_label != 0).                 */
int mpx_scene_cloud(const float *cub_centers, const float *cub_dims, const float *cub_quats, int M1,
                    const float *cyl_centers, const float *cyl_radii, const float *cyl_heights,
                    const float *cyl_quats, int M2, int B, int num_points, uint64_t seed,
                    int64_t env_offset, uint16_t *assign, uint8_t *labels, int32_t *n_obstacles, float *out,
                    int64_t out_batch_stride, int out_point_stride, int write_label,
                    mpx_stream_t stream);

/* ---- PointNet++ set abstraction: pointnet2_ops call sites model.py:27,366-383 ------------- */

/* furthest_point_sample (+ gather_operation): xyz rows at xyz + (b*N + k)*stride, first three
 * floats used.  idx int32 [B,npoint]; new_xyz (optional) rows at
 * new_xyz + (b*npoint + j)*new_stride.  Index semantics follow pointnet2_ops v3.2.0 exactly
 * (start 0, float |p|^2 <= the double 1e-3 skipped, block-size-dependent tie order).  N <= 8192.  */
int mpx_fps(const float *xyz, int B, int N, int stride, int npoint, int32_t *idx, float *new_xyz,
            int new_stride, mpx_stream_t stream);

/* Verification hook (host call, process-wide): which kernel family serves mpx_fps / mpx_ball_query.  value 1 (default) =
 * the fast kernels (one wave per small cloud / Morton-culled FPS; wave-per-query / bucketed ball query); value 0 = the
 * plain kernels they are proven against.  Indices are identical bit for bit either way; the switch lets a test run
 * both on the same clouds in one process (tests/test_gpu_soak.py).  Returns non-zero on a bad selector / value.      */
#define MPX_VARIANT_FPS 0
#define MPX_VARIANT_BALL_QUERY 1
/* 1 (default): a stream's persistent grouped-MLP launches take a private set of unit-queue counters (256 distinct
 * (device, stream) handles per process).  0: handles not seen before get none, as if all 256 were taken -- the fp32
 * launchers then run one unit per wave without a queue (same results), mpx_sa_mlp_bf16x3_factored fails with a message. */
#define MPX_VARIANT_UNIT_QUEUE 2
#define MPX_VARIANT_COUNT_ 3
int mpx_set_variant(int what, int value);
int mpx_get_variant(int what); /* -1: unknown selector */

/* ball_query: first `nsample` indices (ascending) with d2 < radius^2, padded with the first
 * hit, zero when there is none.  idx int32 [B,npoint,nsample].  cnt (optional, int32
 * [B,npoint]) receives the number of real hits (<= nsample): slots [cnt, nsample) are padding. */
int mpx_ball_query(const float *new_xyz, int new_stride, const float *xyz, int stride, int B,
                   int N, int npoint, float radius, int nsample, int32_t *idx, int32_t *cnt,
                   mpx_stream_t stream);

/* The same search, writing the HIT slots only: idx[b, j, 0 .. max(cnt, 1)) (an empty row still gets its slot 0 = 0); the
 * padding slots are left untouched.  For consumers that take `cnt` and never look past it (mpx_sa_mlp / _factored /
 *  _bf16x3 with counts): most of a row is padding, so most of the index writes go away.  cnt is required; nsample <= 256
 * (the most slots per neighbourhood for which the grouped-MLP kernels honour the counts).                            */
int mpx_ball_query_hits(const float *new_xyz, int new_stride, const float *xyz, int stride, int B,
                        int N, int npoint, float radius, int nsample, int32_t *idx, int32_t *cnt,
                        mpx_stream_t stream);

/* order[i] = query ids (0..n-1) sorted by DEcreasing number of rows they contribute to the packed
 * SA kernels, 4*ceil(clamp(cnt,1,nsample)/4) (ties in unspecified order).  scratch: int32[128]
 * device words (zeroed by the call).  Gives the lockstep waves of mpx_sa_mlp_bf16x3 equal work. */
int mpx_sort_queries(const int32_t *cnt, int64_t n, int nsample, int32_t *order, int32_t *scratch,
                     mpx_stream_t stream);

/* QueryAndGroup (use_xyz=True) materialised like the reference does:
 * out [B, 3+C, npoint, nsample]; feat point-major rows at feat + (b*N+k)*feat_stride.       */
int mpx_group_points(const float *xyz, int stride, const float *new_xyz, int new_stride,
                     const float *feat, int feat_stride, int C, const int32_t *idx, int B, int N,
                     int npoint, int nsample, float *out, mpx_stream_t stream);

/* Fused QueryAndGroup + shared MLP (3 x [1x1 conv + ReLU]) + max-pool over the neighbourhood
 * (PointnetSAModule.forward body), fp32 MFMA, nothing materialised in HBM.
 * Channel order of the MLP input is [dx,dy,dz, feat...] like the reference.
 * wpack: weights + biases packed by mpx_sa_pack_weights for this (C, c1, c2, c3).
 * out rows at out + (b*npoint + j)*out_stride, c3 floats each.
 * Supported: (C,c1,c2,c3) = (1,64,64,64) and (64,128,128,256); nsample % 32 == 0.
 * cnt (optional, from mpx_ball_query): only the DISTINCT neighbours are evaluated -- slots
 * [cnt, nsample) repeat the first neighbour, the MLP is per point and max-pooling is idempotent,
 * so the output is bit-identical to walking all nsample slots (cnt == NULL).  Several queries
 * share a wave and their rows are packed (rounded up to 2 rows per query for the narrow module, 4 for the
 * wide one) into the 32-row MFMA tiles.  Large launches with cnt are PERSISTENT: one workgroup per wave slot
 * of the chip takes units of consecutive queries from the device-side queue described at
 * mpx_sa_mlp_bf16x3_factored_wants_order below (same per-(device, stream) counters, same hipGraph note).
 * append_centre != 0 (needs cnt, out_stride >= c3 + 4): columns [c3, c3+3) of every output row also receive the
 * query point's coordinates and column c3+3 a zero -- the rows are then the operand [f | xyz | 0] of the next
 * module's per-point first-layer GEMM (mpx_sa_mlp_factored), model.py:404-407's torch.cat without a second pass. */
int mpx_sa_mlp(const float *xyz, int stride, const float *new_xyz, int new_stride,
               const float *feat, int feat_stride, int C, const int32_t *idx, const int32_t *cnt,
               int B, int N, int npoint, int nsample, const float *wpack, int c1, int c2, int c3,
               float *out, int out_stride, int append_centre, mpx_stream_t stream);
/* Same module with the first layer factored out of the per-(query, neighbour) work:
 *   W1.[p_j - c_i ; f_j] + b1 = pre[j] - ctr[i],  pre = [p ; f].W1^T  (one row per POINT, [B*N, c1]),
 *                                                 ctr = c.W1x^T - b1  (one row per QUERY, [B*npoint, c1]),
 * both plain mpx_linear calls by the caller.  The kernel gathers pre rows, subtracts the query row,
 * applies ReLU and continues with layers 2-3 and the max-pool as above (wpack unchanged; its layer-1
 * block is not read).  15 % less matrix work for the (64,128,128,256) module; equal to mpx_sa_mlp
 * up to the rounding of that one re-associated sum.  cnt is required.  (Persistent for large launches, as above.) */
int mpx_sa_mlp_factored(const float *pre, const float *ctr, const int32_t *idx, const int32_t *cnt, int B,
                        int N, int npoint, int nsample, const float *wpack, int C, int c1, int c2, int c3,
                        float *out, int out_stride, mpx_stream_t stream);
/* mpx_sa_mlp_factored on the bf16 matrix cores (split products, see mpx_sa_mlp_bf16x3): wpack from
 * mpx_sa_pack_bf16x3 (its layer-1 blocks are not read); order (optional) from mpx_sort_queries.
 * nsample <= 128 (the kernel's row -> query table is sized for it; larger neighbourhoods are refused). */
/* host query: does mpx_sa_mlp_bf16x3_factored use `order`?  Always 0 since version 300: its one kernel is persistent
 * and takes units of 8 queries from a device-side queue (`order` is ignored; pass NULL).  The queue is 32 bytes of
 * library-image memory per (device, stream), zeroed on the launch stream in front of the kernel: concurrent calls on
 * DIFFERENT streams never share counters (up to 256 streams per process).  hipGraph: a captured call bakes its capture
 * stream's queue in -- do not replay two graphs that were captured on the same stream concurrently on two streams.   */
int mpx_sa_mlp_bf16x3_factored_wants_order(void);
int mpx_sa_mlp_bf16x3_factored(const float *pre, const float *ctr, const int32_t *idx, const int32_t *cnt,
                               const int32_t *order, int B, int N, int npoint, int nsample, const void *wpack,
                               int C, int c1, int c2, int c3, float *out, int out_stride, mpx_stream_t stream);
/* number of floats mpx_sa_pack_weights writes for this configuration (host call)            */
int64_t mpx_sa_pack_size(int C, int c1, int c2, int c3);
/* w1 [c1,3+C], w2 [c2,c1], w3 [c3,c2] row-major (Conv2d 1x1 weights), b* biases -> wpack    */
int mpx_sa_pack_weights(const float *w1, const float *b1, const float *w2, const float *b2,
                        const float *w3, const float *b3, int C, int c1, int c2, int c3,
                        float *wpack, mpx_stream_t stream);

/* Split-bf16 ("bf16x3") variant of mpx_sa_mlp: every fp32 product is evaluated as
 * x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on the bf16 matrix cores with fp32 accumulation (5.3x fewer
 * MFMA cycles; |error| ~ 2^-16 relative per product, ~3e-7 on the policy output).  Same arguments
 * and outputs as mpx_sa_mlp; wpack comes from mpx_sa_pack_bf16x3 (size in BYTES from
 * mpx_sa_pack_bf16x3_size).  Opt-in: the fp32 kernel is the parity default.  cnt as in mpx_sa_mlp
 * (distinct neighbours only, rows packed; NULL = all slots).  `order` (optional, from
 * mpx_sort_queries) is the order in which waves take the queries: the 8 waves of a workgroup walk
 * the weight stream in lockstep and run max(tiles) of their row counts.  A module whose whole pack fits
 * LDS (the first one, 1+3 -> 64 -> 64 -> 64) runs a weight-resident kernel instead that takes the queries
 * in their natural order (up to nsample = 128): mpx_sa_mlp_bf16x3_wants_order (host query; the launcher uses the
 * same predicate) says whether `order` is used for a module and neighbourhood size.  append_centre as in mpx_sa_mlp
 * (only where `order` is not used).                                                                        */
int mpx_sa_mlp_bf16x3_wants_order(int C, int c1, int c2, int c3, int nsample);
int mpx_sa_mlp_bf16x3(const float *xyz, int stride, const float *new_xyz, int new_stride,
                      const float *feat, int feat_stride, int C, const int32_t *idx,
                      const int32_t *cnt, const int32_t *order, int B, int N, int npoint,
                      int nsample, const void *wpack, int c1, int c2, int c3, float *out,
                      int out_stride, int append_centre, mpx_stream_t stream);
int64_t mpx_sa_pack_bf16x3_size(int C, int c1, int c2, int c3);
int mpx_sa_pack_bf16x3(const float *w1, const float *b1, const float *w2, const float *b2,
                       const float *w3, const float *b3, int C, int c1, int c2, int c3, void *wpack,
                       mpx_stream_t stream);

/* The group-all set-abstraction module (PointnetSAModule(mlp=[256(+3), 512, 512, 1024]), npoint = None: model.py:377-383)
 * as ONE kernel: x rows [B*rows, K1] (ldx floats apart; [xyz | features | 0], K1 = 272 = 259 padded to whole 16-float
 * slabs), rows = 128 per environment -> Linear + ReLU (K1 -> c1) -> Linear + ReLU (c1 -> c2) -> Linear + ReLU (c2 -> c3)
 * -> max over the environment's rows -> out [B, c3] (ldo floats apart).  Nothing between the input rows and the pooled row
 * touches HBM (the layer-by-layer form writes and re-reads two [B*128, 512] intermediates).  fp32 MFMA, exact fp32
 * products; the summation order inside a dot product differs from mpx_linear's (1e-7 relative).  One workgroup per
 * environment: meant for B >= 256.  pack: mpx_sa3_pack_weights (size in floats: mpx_sa3_pack_size; -1 = unsupported
 * shape; supported: (K1, c1, c2, c3) = (272, 512, 512, 1024)); w1 is [c1, k1_real] row-major with k1_real <= K1 real
 * input columns (259), w2 [c2, c1], w3 [c3, c2].                                                                       */
int64_t mpx_sa3_pack_size(int K1, int c1, int c2, int c3);
int mpx_sa3_pack_weights(const float *w1, int k1_real, const float *b1, const float *w2, const float *b2, const float *w3,
                         const float *b3, int K1, int c1, int c2, int c3, float *pack, mpx_stream_t stream);
int mpx_sa3_chain(const float *x, int ldx, int B, int rows, const float *pack, int K1, int c1, int c2, int c3, float *out,
                  int ldo, mpx_stream_t stream);

/* ---- dense layers: model.py:47-66, 385-393 ------------------------------------------------- */

#define MPX_ACT_NONE 0
#define MPX_ACT_RELU 1
#define MPX_ACT_LEAKY 2 /* slope 0.01, nn.LeakyReLU() default */

/* y[m, n] = act(sum_k x[m,k] * w[n,k] + bias[n]);  x rows at x + m*ldx, y rows at y + m*ldy.
 * fp32 MFMA.  K % 4 == 0, ldx % 4 == 0, x and w 16-byte aligned.                             */
int mpx_linear(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K,
               int act, float *y, int ldy, mpx_stream_t stream);

/* The same layer with a caller-provided workspace that lets skinny problems (few 128x128 output tiles, long K:
 * the fc / decoder layers of a single-problem or few-hundred-problem rollout) split K over the chip's CUs.
 * mpx_linear_workspace(M,N,K) = bytes this shape wants (0: it would not split); the partial sums are added in
 * slice order, so results are deterministic (they differ from mpx_linear's by fp32 summation order only).
 * workspace == NULL behaves exactly like mpx_linear.                                          */
int64_t mpx_linear_workspace(int M, int N, int K);
int mpx_linear_ws(const float *x, int ldx, const float *w, const float *bias, int M, int N, int K,
                  int act, float *y, int ldy, void *workspace, int64_t workspace_bytes,
                  mpx_stream_t stream);

/* Last layer of the group-all SA module with its max-pool fused (model.py:383):
 *   y[g, n] = max over the `rows` rows of group g of relu(x[g*rows + r, :] . w[n, :] + bias[n])
 * rows must be 128 (one workgroup tile) and divide M; y [M/rows, N] is written (zeroed first).  */
int mpx_linear_rowmax(const float *x, int ldx, const float *w, const float *bias, int M, int N,
                      int K, int rows, float *y, int ldy, mpx_stream_t stream);

/* nn.GroupNorm(groups, C) (eps 1e-5, biased variance) followed by LeakyReLU(0.01), in place
 * allowed.  x,y [M,C].                                                                       */
int mpx_groupnorm_leaky(const float *x, const float *gamma, const float *beta, int M, int C,
                        int groups, float eps, float *y, mpx_stream_t stream);
/* the same, with the result written in the bf16x3 dense layers' pairs form (see mpx_split_bf16; C a multiple of
 * 16, ldp >= 2 C bf16 elements) instead of fp32 rows: the operand of the dense layer that follows.  */
int mpx_groupnorm_leaky_to_pairs(const float *x, const float *gamma, const float *beta, int M, int C,
                                 int groups, float eps, void *y_pairs, int ldp, mpx_stream_t stream);

/* max over `rows` consecutive rows: x [G*rows, C] -> y [G, C]  (max_pool2d of the group-all
 * SA module, model.py:383)                                                                   */
int mpx_rowmax(const float *x, int ldx, int G, int rows, int C, float *y, int ldy,
               mpx_stream_t stream);

/* rows[i, col0 : col0+ncols] = src[i, :ncols] and rows[i, col0+ncols : col0+ncols+nzero] = 0 for i < n (strides in
 * floats).  Builds SA2's per-point operand rows [f1 | xyz1 | 0] next to the features SA1 wrote (model.py:404-407's
 * torch.cat of xyz and features, without the copy of the features).                                              */
int mpx_append_columns(const float *src, int src_stride, int ncols, int nzero, int64_t n, float *rows,
                       int row_stride, int col0, mpx_stream_t stream);

/* ---- the whole policy forward in one call: MotionPolicyNetwork.forward, model.py:75-91 ------------------------
 * For callers without Python (a native planning node): slab + joint configuration in, displacement out.  Host-side
 * orchestration of the kernels above in the order and shapes of mpinets_amd/model.py, whose fp32 output it
 * reproduces bit for bit; the caller's stream (for B <= 512 also an internal second stream, forked and joined with
 * events -- hipGraph-capturable; concurrent calls from several host threads on one device are safe: the internal
 * stream is held exclusively from fork to join), no allocation, no synchronisation.
 *
 * Weights: device pointers, fp32, [out, in] row-major like nn.Linear / 1x1 Conv2d, prepared once:
 *   sa1_pack, sa2_pack   mpx_sa_pack_weights of SA_modules.{0,1} (C = 1 and 64)
 *   sa2_wpoint [128,68]  SA2's first layer over rows [f1 (64) | xyz (3) | 0]   (W1[:,3:] | W1[:,:3] | 0)
 *   sa2_wcentre [128,4]  ... over rows [xyz (3) | *]                             (W1[:,:3] | 0)
 *   sa2_nb1 [128]        minus its bias
 *   sa3_w[0] [512,272]   group-all first layer, K padded 259 -> 272 with zero columns; sa3_w[1] [512,512];
 *                        sa3_w[2] [1024,512]; sa3_b[i] the biases (the layer-by-layer form: B < 256 problems)
 *   sa3_pack             the same three layers packed by mpx_sa3_pack_weights (the fused form: B >= 256); NULL: the
 *                        layer-by-layer form at every batch size (same results within rounding)
 *   fc_w / fc_b          1024 -> 4096 -> 2048 -> 2048; gn_g / gn_b: the two GroupNorm(16) affine vectors
 *   qe_w[0] [32,8]       joint encoder, first layer K padded 7 -> 8; then 32 -> 64 -> 128 -> 128 -> 64
 *   de_w / de_b          decoder 2112 -> 512 -> 256 -> 128 -> 7                                                  */
typedef struct mpx_policy_weights {
  const float *sa1_pack, *sa2_pack, *sa2_wpoint, *sa2_wcentre, *sa2_nb1;
  const float *sa3_w[3], *sa3_b[3];
  const float *fc_w[3], *fc_b[3], *gn_g[2], *gn_b[2];
  const float *qe_w[5], *qe_b[5];
  const float *de_w[4], *de_b[4];
  const float *sa3_pack; /* mpx_sa3_pack_weights of the group-all module (K1 = 272): the fused chain of B >= 256 problems, or NULL */
} mpx_policy_weights;

/* bytes of 256-byte aligned device workspace a batch of B problems with N-point slabs needs */
int64_t mpx_policy_workspace(int B, int N);
/* xyz [B,N,4] (x, y, z, label; N in [512, 8192]), q [B,7] normalised to [-1,1] -> dq [B,7] (normalised space).
 * B <= 65535.                                                                                                    */
int mpx_policy_forward(const mpx_policy_weights *w, const float *xyz, int N, const float *q, int B,
                       float *dq, void *workspace, int64_t workspace_bytes, mpx_stream_t stream);

/* Per-call robot-point subset.  robofin's FrankaSampler.sample redraws np.random.choice(P, num_points, replace=False) on
 * EVERY call, ONE subset shared by the batch (call sites mpinets/model.py:170-181, run_inference.py:188-189); NumPy's
 * global RNG cannot be reproduced on a device, so the distribution is what is kept: out int32 [n_out] <- n_out of the
 * `total` table rows, uniformly without replacement, in uniform order (row i keyed by Philox4x32-10 with counter
 * (i >> 2, draw, 11, 0) and key `seed`; the n_out smallest (key, row) pairs in that order).  n_out <= 4096.  Depends on
 * (seed, draw) only: every rank of a sharded batch draws the same subset, like one process would.                 */
int mpx_draw_subset(int total, int n_out, uint64_t seed, int draw, int32_t *out, mpx_stream_t stream);

/* ---- closed-loop steps in one call: TrainingMotionPolicyNetwork.rollout's loop body (model.py:160-181) plus the
 * collision check of validation_step (model.py:293-314), i.e. what mpinets_amd.rollout.RolloutEngine.step() does
 * with a static scene:  dq = policy(xyz, q_norm);  q_norm = clamp(q_norm + dq, -1, 1);  q = unnormalise(q_norm);
 * robot rows of the slab <- FK cloud of q;  flags = any collision sphere of q inside a primitive.
 * Everything besides the policy weights that the step reads (device pointers, shared by the batch unless noted): */
typedef struct mpx_rollout_scene {
  const float *limits;           /* [7,2] joint limits (lower, upper)                                            */
  float finger;                  /* prismatic finger opening used for FK                                        */
  const float *table_pts;        /* FrankaSampler point table [P,3] in link frames ...                         */
  const int32_t *table_link;     /* ... and the link of each point [P]                                         */
  const int32_t *subset;         /* the n_robot table rows written to the slab (NULL: the first n_robot)       */
  int n_robot;                   /* robot rows at the head of every slab (2048)                                */
  const float *sph_centers, *sph_radii; /* collision spheres [S,3], [S] in link frames ...                     */
  const int32_t *sph_link;       /* ... and their links [S]                                                    */
  int n_spheres;
  const float *cub_frames, *cub_dims;   /* per environment: [B,M1,4,4] inverse frames (mpx_prim_frames), [B,M1,3] */
  int M1;
  const float *cyl_frames, *cyl_radii, *cyl_heights;   /* [B,M2,4,4], [B,M2], [B,M2]                           */
  int M2;
} mpx_rollout_scene;

/* What a multi-step rollout does besides the plain step (all optional; a zeroed struct with steps = 1 is
 * mpx_rollout_step): the engine-side form of rollout_until_success's loop body (run_inference.py:171-189) and of
 * the "closed-loop point-cloud re-render" workload.                                                              */
typedef struct mpx_rollout_options {
  int steps;                      /* closed-loop steps to enqueue (no host synchronisation in between)            */
  int first_step;                 /* index of the first of them: the re-render seed schedule continues an earlier call */
  /* scene re-render at the start of every step (n_scene = 0: the scene rows of the slab stay as they are):
   * n_scene points per environment drawn from the scene's primitives by mpx_scene_cloud with seed
   * scene_seed + 7919 * step and global environment ids env_offset + row, written to slab rows
   * [n_robot, n_robot + n_scene).  Needs the primitives' raw poses beside the frames in mpx_rollout_scene.      */
  int n_scene;
  uint64_t scene_seed;
  int64_t env_offset;
  const float *cub_centers, *cub_quats;   /* [B,M1,3], [B,M1,4] (w,x,y,z) */
  const float *cyl_centers, *cyl_quats;   /* [B,M2,3], [B,M2,4]           */
  /* success tracking (target_poses = NULL: off): after every joint update, environments whose right_gripper is
   * within pos_tol metres and cos_rot_tol (cosine of the angle) of target_poses [B,4,4] set done[b] (int32 [B],
   * zeroed by the caller before the first call) and keep their configuration from then on; steps_taken (optional,
   * int32 [B], zeroed by the caller) counts the policy steps each environment took until it was done.          */
  const float *target_poses;
  float pos_tol, cos_rot_tol;
  int32_t *done, *steps_taken;
  /* optional [B, trajectory_len, 7]: waypoint trajectory_row + i of every environment receives q after step i of
   * this call (row 0 is normally the caller's start configuration, so the first call passes trajectory_row = 1)  */
  float *trajectory;
  int trajectory_len, trajectory_row;
  /* per-step robot-point subset (subset_table_size = 0: off, the scene's fixed `subset` is used): before the FK cloud
   * refresh of step s, subset_buf (int32 [n_robot], caller's device buffer) <- mpx_draw_subset(subset_table_size,
   * n_robot, subset_seed, s) and the refresh writes THOSE table rows -- the reference's per-call redraw.         */
  int subset_table_size;
  uint64_t subset_seed;
  int32_t *subset_buf;
} mpx_rollout_options;

/* workspace bytes (256-byte aligned) of mpx_rollout / mpx_rollout_step */
int64_t mpx_rollout_workspace(int B, int N);
/* xyz [B,N,4] slab (robot rows -- and scene rows when re-rendering -- rewritten in place), q_norm [B,7] in/out,
 * q [B,7] out (radians), flags int32 [B] OR-ed into (non-zero: some step's configuration was in collision),
 * min_sdf [B, n_spheres] out or NULL (last step's).  Bit-identical to RolloutEngine.step() called `steps` times.  */
int mpx_rollout(const mpx_policy_weights *w, const mpx_rollout_scene *scene, const mpx_rollout_options *options,
                float *xyz, int N, float *q_norm, float *q, int B, int32_t *flags, float *min_sdf, void *workspace,
                int64_t workspace_bytes, mpx_stream_t stream);
/* one plain step (static scene, no success tracking): mpx_rollout with a zeroed options struct and steps = 1 */
int mpx_rollout_step(const mpx_policy_weights *w, const mpx_rollout_scene *scene, float *xyz, int N,
                     float *q_norm, float *q, int B, int32_t *flags, float *min_sdf, void *workspace,
                     int64_t workspace_bytes, mpx_stream_t stream);

/* ---- measurement hooks (the scripts under tools/probes; no product path calls them) ------------------------------------------
 * The same launches as mpx_sa3_chain / mpx_sa_mlp_bf16x3_factored through instantiations that write s_memtime stamps
 * of one wave at the kernel's phase boundaries.                                                                    */
/* probe: int64 [>= 32]; B >= 301 (workgroup 300's wave 0 is the one stamped) */
int mpx_sa3_chain_probe(const float *x, int ldx, int B, const float *pack, float *out, int ldo, int64_t *probe,
                        mpx_stream_t stream);
/* probe: int64 [>= 128] or NULL (off): while set, every mpx_sa_mlp_bf16x3_factored call first runs the stamped
 * instantiation (process-wide switch, not thread-safe) */
int mpx_sa2_bf16x3_set_probe(int64_t *probe);
/* mpx_sa3_front_bf16x3 with stamps (probe: int64 [>= 64]; B > 300) */
int mpx_sa3_front_bf16x3_probe(const float *x, int ldx, int B, const void *pack, void *y_pairs, int ldp, int64_t *probe,
                               mpx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPINETS_HIP_H */
