/*
 * vitta_hip.h — C ABI of libvitta_hip.so (gfx950 / MI355X).
 *
 * The reference (wlin-at/ViTTA) has no FFI layer: its hot path is a chain of stock
 * PyTorch ops fired from forward hooks.  This header is the boundary one level
 * below the reference's Python operator protocol; every entry point names the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer named d_* is a caller-owned DEVICE pointer (tensor.data_ptr());
 *     pointers named h_* are host pointers read synchronously during the call;
 *   - every launch function takes an explicit hipStream_t (passed as void*) and is
 *     asynchronous; nothing synchronises the device;
 *   - return value: 0 = ok, <0 = error (see vitta_status_string); never throws,
 *     never aborts; no global mutable state, re-entrant;
 *   - all arithmetic is fp32 (the reference is fp32 end to end), partial moments
 *     are merged with Chan's formula, the tiny cross-block combines run in fp64.
 */
#ifndef VITTA_HIP_H_
#define VITTA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITTA_ABI_VERSION 1

/* status codes */
#define VITTA_OK 0
#define VITTA_ERR_INVALID_ARG (-1)
#define VITTA_ERR_LAUNCH (-2)
#define VITTA_ERR_ALLOC (-3)
#define VITTA_ERR_UNSUPPORTED (-4)
#define VITTA_ERR_WORKSPACE (-5)

/* feature layouts of a hooked norm layer's output */
#define VITTA_LAYOUT_NCHW 0 /* [outer=N*T][C][inner=H*W]   BatchNorm2d output (norm_stats_utils.py:188-193) */
#define VITTA_LAYOUT_NHWC 1 /* [outer=N*T*H*W][C][inner=1] LayerNorm output   (norm_stats_utils.py:222-230) */

/* regularisation types of compute_regularization (norm_stats_utils.py:531-542) */
#define VITTA_REG_L1 0
#define VITTA_REG_MSE 1
#define VITTA_REG_KLD 2

/* maximum number of hooked layers in one plan (pointers travel as kernel arguments) */
#define VITTA_MAX_LAYERS 96

int vitta_abi_version(void);
const char* vitta_status_string(int status);

/* One hooked layer: static shape information. */
typedef struct vitta_layer_shape {
  int64_t outer;  /* NCHW: N*V*T frames ; NHWC: N*V*T'*H*W rows */
  int32_t C;      /* channels */
  int64_t inner;  /* NCHW: H*W ; NHWC: 1 */
  int32_t layout; /* VITTA_LAYOUT_* */
} vitta_layer_shape;

/* --------------------------------------------------------------------------
 * Plan: the static part (shapes -> block tables, packed channel offsets) of the
 * batched multi-layer launches.  Packed per-channel arrays have length
 * vitta_plan_total_channels(); layer l owns [chan_off(l), chan_off(l)+C_l).
 * -------------------------------------------------------------------------- */
typedef struct vitta_plan vitta_plan;

int vitta_plan_create(const vitta_layer_shape* h_shapes, int n_layers, int target_blocks,
                      vitta_plan** out_plan);
/* Same with the frame split of NCHW layers fixed by the caller (h_nsplit[l] > 0; 0 / NULL: the library's choice).
 * A plan whose per-workgroup triples are written by fused BatchNorm passes (vitta_bn_act_fwd_f32, one launch PER
 * LAYER) wants each layer split over enough workgroups to fill the GPU on its own; the batched kernel (ONE launch
 * for all layers) wants few, long walks. */
int vitta_plan_create_split(const vitta_layer_shape* h_shapes, int n_layers, int target_blocks,
                            const int32_t* h_nsplit, vitta_plan** out_plan);
/* The plan itself is a host object.  Its device tables live in a CALLER-OWNED device buffer of
 * vitta_plan_table_bytes() bytes (256-byte aligned) that must outlive the plan's launches;
 * vitta_plan_upload copies them there (stream-ordered) and must precede the first launch.  The
 * library never allocates or frees device memory. */
size_t vitta_plan_table_bytes(const vitta_plan* plan);
int vitta_plan_upload(vitta_plan* plan, void* d_tables, size_t bytes, void* stream);
void vitta_plan_destroy(vitta_plan* plan);
/* tuning knobs of a plan */
#define VITTA_OPT_NT_LOADS 1 /* non-temporal loads in the NCHW moments kernel (value 0/1) */
int vitta_plan_set_option(vitta_plan* plan, int option, int value);
int64_t vitta_plan_total_channels(const vitta_plan* plan);
int64_t vitta_plan_channel_offset(const vitta_plan* plan, int layer);
size_t vitta_plan_workspace_bytes(const vitta_plan* plan);
int64_t vitta_plan_num_blocks(const vitta_plan* plan); /* workgroups of the moments launch (both layouts) */

/* --------------------------------------------------------------------------
 * A1/A2 — per-channel spatio-temporal moments over (N*V, T, H, W).
 * Replaces CombineNormStatsRegHook_onereg.compute_reg_for_NCTHW's
 *   permute -> contiguous -> mean((0,2,3,4)) -> permute -> contiguous -> var(1, unbiased=False)
 * (utils/norm_stats_utils.py:238-243, BN2d branch :188-204, LN branch :222-230) and
 * ComputeNormStatsHook.compute_stat_for_NCTHW (:92-95).
 *
 * Batched form: one launch per layout over all layers of the plan.
 *   h_x[l]    device pointer of layer l's feature (fp32, contiguous in its layout)
 *   d_shift   [total_channels] or NULL: per-channel shift k_c (use the source mean);
 *   d_cnt     [n_layers]        out: element count per channel n_l (as float)
 *   d_s1      [total_channels]  out: sum_i (x_i - k_c)
 *   d_s2      [total_channels]  out: sum_i (x_i - k_c)^2
 * (cnt,s1,s2) are ADDITIVE across data-parallel ranks: one SUM all-reduce over the
 * three arrays gives the moments of the pooled batch.
 * -------------------------------------------------------------------------- */
int vitta_moments_batched_f32(const vitta_plan* plan, const void* const* h_x, const float* d_shift,
                              float* d_cnt, float* d_s1, float* d_s2, void* d_workspace,
                              size_t workspace_bytes, void* stream);

/* The two stages of vitta_moments_batched_f32 as separate calls (same arguments): the streaming
 * pass over the features (per-workgroup partial triples into the workspace) and the tiny fp64
 * combine.  bench.py brackets the first with events to time the HBM-bound kernel alone. */
int vitta_moments_partials_f32(const vitta_plan* plan, const void* const* h_x, void* d_workspace,
                               size_t workspace_bytes, void* stream);
/* Measurement aid (bench.py's live roofline figure): the same first-stage launch with two hipEvents attached to the
 * kernel's DISPATCH (hipExtLaunchKernelGGL), so that elapsed(start, stop) is the kernel's own duration -- what
 * rocprofv3 --kernel-trace reports -- without the barrier packets and launch latency an event pair recorded
 * around a lone launch also brackets.  Events come from vitta_event_create (plain hipEvent_t handles). */
int vitta_moments_partials_timed_f32(const vitta_plan* plan, const void* const* h_x, void* d_ws, size_t ws_bytes,
                                     void* stream, void* ev_start, void* ev_stop);
int vitta_event_create(void** out_event);
void vitta_event_destroy(void* event);
int vitta_event_elapsed_ms(void* ev_start, void* ev_stop, float* out_ms); /* waits for ev_stop */
int vitta_moments_finalize_f32(const vitta_plan* plan, const float* d_shift, float* d_cnt, float* d_s1,
                               float* d_s2, const void* d_workspace, size_t workspace_bytes, void* stream);

/* bfloat16 features ("_bf16 input variants with fp32 accumulate"; 8d: halves the algorithmic bytes of a
 * bf16 activation pipeline, 507 MB instead of 1 015 MB per video at C5): h_x[l] points at 2-byte elements in the same
 * layouts, widened to fp32 in registers; partial triples, sums and the finalize stay fp32 / fp64 exactly as above, so
 * the result equals the fp32 path run on the widened values.  Same plan; a layer whose plane (NCHW) or channel count
 * (NHWC) is a multiple of four needs 8-byte aligned features. */
int vitta_moments_batched_bf16(const vitta_plan* plan, const void* const* h_x, const float* d_shift,
                               float* d_cnt, float* d_s1, float* d_s2, void* d_workspace,
                               size_t workspace_bytes, void* stream);
int vitta_moments_partials_bf16(const vitta_plan* plan, const void* const* h_x, void* d_workspace,
                                size_t workspace_bytes, void* stream);

/* Convert additive sums to (mean, biased var): mean = k + s1/n, var = s2/n - (s1/n)^2. */
int vitta_moments_to_meanvar_f32(const vitta_plan* plan, const float* d_shift, const float* d_cnt,
                                 const float* d_s1, const float* d_s2, float* d_mean, float* d_var,
                                 void* stream);

/* Single-layer conveniences (SURVEY 8b names).  d_mean/d_var: [C].  Workspace from
 * vitta_moments_workspace_bytes(outer, C, inner, layout). */
size_t vitta_moments_workspace_bytes(int64_t outer, int32_t C, int64_t inner, int32_t layout);
int vitta_moments_nchw_f32(const float* d_x, int64_t NT, int32_t C, int64_t HW, float* d_mean,
                           float* d_var, void* d_workspace, size_t workspace_bytes, void* stream);
int vitta_moments_nhwc_f32(const float* d_x, int64_t rows, int32_t C, float* d_mean, float* d_var,
                           void* d_workspace, size_t workspace_bytes, void* stream);
int vitta_moments_nchw_bf16(const uint16_t* d_x, int64_t NT, int32_t C, int64_t HW, float* d_mean,
                            float* d_var, void* d_workspace, size_t workspace_bytes, void* stream);
int vitta_moments_nhwc_bf16(const uint16_t* d_x, int64_t rows, int32_t C, float* d_mean, float* d_var,
                            void* d_workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------------
 * A3 + A4 — EMA update and alignment loss, all layers in one launch.
 * Replaces MovingAverageTensor.update (utils/utils_.py:204-211; avg0 = 0, no bias
 * correction) and compute_regularization / compute_kld
 * (utils/norm_stats_utils.py:531-542, :8-16), plus the autograd bookkeeping of A6:
 *   d_ema_mean/d_ema_var [total_channels] in/out  EMA state (zero-initialised)
 *   d_src_mean/d_src_var [total_channels]         source statistics
 *   momentum                                       args.momentum_mvg
 *   d_layer_loss [n_layers] out  r_feature of every hook
 *   d_total_loss [1]        out  sum over layers (loss_reg, corpus/basics.py:658-661)
 *   d_mu   [total_channels] out  batch mean mu_c (needed by the backward)
 *   d_coef_a,d_coef_b       out  dL/dx[i,c] = a_c + b_c * (x[i,c] - mu_c)
 *   d_zero_word [1] or NULL out  one more device float set to 0 by the launch (the engine's gradient scale, reset once per step)
 * ONE launch (round 5): a workgroup per layer, the last to arrive (counter in the plan's uploaded tables, zero at rest) adds the
 * layers in layer order.  Not re-entrant per plan: one alignment launch of a plan at a time (they are issued on one stream).
 * -------------------------------------------------------------------------- */
int vitta_stat_align_fwd_f32(const vitta_plan* plan, const float* d_shift, const float* d_cnt,
                             const float* d_s1, const float* d_s2, float* d_ema_mean,
                             float* d_ema_var, const float* d_src_mean, const float* d_src_var,
                             float momentum, int reg_type, float* d_layer_loss, float* d_total_loss,
                             float* d_mu, float* d_coef_a, float* d_coef_b, float* d_zero_word, void* stream);

/* --------------------------------------------------------------------------
 * A6 — backward of A1-A4 w.r.t. the hooked feature:
 *   gin[i,c] = gout[i,c] + gscale * (a_c + b_c * (x[i,c] - mu_c))
 * d_gout may be NULL (treated as 0); d_gin may alias d_gout.  d_gscale is a DEVICE
 * scalar (upstream gradient of loss_reg, e.g. lambda_feature_reg) or NULL (= 1).
 * d_mu/d_coef_a/d_coef_b point at the layer's [C] slice.
 * -------------------------------------------------------------------------- */
int vitta_stat_align_bwd_f32(const float* d_x, const float* d_gout, float* d_gin, int64_t outer,
                             int32_t C, int64_t inner, int32_t layout, const float* d_mu,
                             const float* d_coef_a, const float* d_coef_b, const float* d_gscale,
                             void* stream);

/* --------------------------------------------------------------------------
 * A5 — prediction-consistency loss and its gradient.
 * Replaces compute_pred_consis (utils/pred_consistency_utils.py:15-31):
 *   p_v = softmax(logits[:,v,:]); pbar = mean_v p_v (not detached);
 *   loss = sum_v sum_{b,k} |p_v - pbar| / V.
 * d_logits [B,V,K] contiguous; d_loss [1 + B]: [0] = loss, [1..B] = per-video partial sums
 * (scratch); d_grad [B,V,K] = dloss/dlogits (may be NULL).  2*V*K*4 bytes must fit 48 KB of LDS.
 * -------------------------------------------------------------------------- */
int vitta_pred_consis_f32(const float* d_logits, int32_t B, int32_t V, int32_t K, float* d_loss,
                          float* d_grad, void* stream);

/* --------------------------------------------------------------------------
 * A9 — TAM (temporal adaptive module) aggregation, fused.
 * Replaces the tail of TAM.forward (models/tanet_models/temporal_module.py:47-65):
 *   y = gate * x ; out[n,t,c,:] = sum_{j=0..2} K[n,c,j] * y[n,t+j-1,c,:]  (zero pad in T)
 * on x [N*T, C, H*W] WITHOUT the two permute+contiguous copies.
 *   d_x    [N*T, C, HW]   d_gate [N, C, T]   d_kern [N*C, 3]   d_out [N*T, C, HW]
 * and of F.adaptive_avg_pool2d on the permuted copy (:51):
 *   d_pool [N, C, T] = mean over HW.
 * Backward: d_gx (w.r.t. x), d_ggate, d_gkern from d_gout.  d_ggate must have room for
 * N*C*T*4 floats: the [N,C,T] result followed by an [N,C,T,3] scratch of row dot products.
 * vitta_tam_pool_bwd adds gpool[n,c,t]/HW to every element of row (n,t,c) of d_gx_accum.
 * -------------------------------------------------------------------------- */
int vitta_tam_pool_f32(const float* d_x, int32_t N, int32_t T, int32_t C, int32_t HW,
                       float* d_pool, void* stream);
int vitta_tam_agg_fwd_f32(const float* d_x, const float* d_gate, const float* d_kern, int32_t N,
                          int32_t T, int32_t C, int32_t HW, float* d_out, void* stream);
int vitta_tam_agg_bwd_f32(const float* d_x, const float* d_gate, const float* d_kern,
                          const float* d_gout, int32_t N, int32_t T, int32_t C, int32_t HW,
                          float* d_gx, float* d_ggate, float* d_gkern, void* stream);
int vitta_tam_pool_bwd_f32(const float* d_gpool, int32_t N, int32_t T, int32_t C, int32_t HW,
                           float* d_gx_accum, void* stream);

/* --------------------------------------------------------------------------
 * A10 -- fused 3-D (shifted-)window multi-head self-attention of Video Swin.
 * Replaces WindowAttention3D.forward between the qkv and proj Linears
 * (models/videoswintransformer_models/swin_transformer.py:144-168): q*scale, q@k^T, + relative
 * position bias, + shift mask, softmax, @v and the head reshapes/permutes, per window and head,
 * without materialising the [N x N] attention matrix.
 *   d_qkv  [B_, N, 3, nH, head_dim]  output of the qkv Linear, untouched
 *   d_bias [nH, N, N]                relative_position_bias_table[relative_position_index]
 *   d_mask [nW, N, N] or NULL        window b uses mask[b % nW] (B_ % nW == 0)
 *   d_out  [B_, N, nH*head_dim]      input layout of the proj Linear
 *   d_lse  [B_, nH, N]               row max + log(row sum), consumed by the backward
 * Backward: d_dqkv [B_, N, 3, nH, head_dim] (fully written), d_dbias [nH, N, N] or NULL
 * (accumulated with atomics: zero it first), d_delta [B_, nH, N] scratch.
 * Supported: head_dim == 32, N <= 400 (vitta_wmsa_supported); otherwise VITTA_ERR_UNSUPPORTED.
 * -------------------------------------------------------------------------- */
int vitta_wmsa_supported(int32_t N, int32_t head_dim);
int vitta_wmsa_fwd_f32(const float* d_qkv, const float* d_bias, const float* d_mask, int32_t nW, int64_t B_,
                       int32_t N, int32_t nH, int32_t head_dim, float scale, float* d_out, float* d_lse,
                       void* stream);
int vitta_wmsa_bwd_f32(const float* d_qkv, const float* d_bias, const float* d_mask, int32_t nW, int64_t B_,
                       int32_t N, int32_t nH, int32_t head_dim, float scale, const float* d_out,
                       const float* d_dout, const float* d_lse, float* d_delta, float* d_dqkv, float* d_dbias,
                       void* stream);

/* TAM branches fused: pooled [N,C,T] -> kern [N*C,3] (G: Linear-BN1d-ReLU-Linear-Softmax) and gate [N,C,T]
 * (L: Conv1d k3-BN1d-ReLU-Conv1d k1-Sigmoid), temporal_module.py:27-41,53-55, every BatchNorm1d in eval().
 * One launch forward, one backward (the module chain is ~14 + ~25 launches of KB-sized tensors).
 *   h_bn_g / h_bn_l: HOST arrays of 4 device pointers {weight, bias, running_mean, running_var};
 *   d_hpre [2, N, C/4, T]: conv1 output before BN and after BN+ReLU, saved for the backward;
 *   d_gpooled: N*C*T floats of result followed by N*(C/4)*T floats of scratch;
 *   backward: h_dbn = {dG.weight, dG.bias, dL.weight, dL.bias} and h_dw = {dG.0.w, dG.3.w, dL.0.w, dL.3.w}
 *   (entries or the whole array may be NULL when the weights are frozen) are ACCUMULATED: zero them first.
 *   pooled_tc: 0 = d_pooled is float [N, C, T]; 1 = int64 fixed point (32 fractional bits) [N, T, C] -- frame-major, what a
 *   convolution's VITTA_CONV_POOL epilogue accumulates.
 * Supported: T <= 16, C % 4 == 0 (vitta_tam_branch_supported). */
int vitta_tam_branch_supported(int32_t C, int32_t T);
int vitta_tam_branch_fwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, float* d_kern, float* d_gate,
                             float* d_hpre, int32_t pooled_tc, void* stream);
int vitta_tam_branch_bwd_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                             const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                             const float* d_w3, int32_t N, int32_t C, int32_t T, const float* d_kern,
                             const float* d_gate, const float* d_hpre, const float* d_gkern, const float* d_ggate,
                             float* d_gpooled, float* const* h_dbn, float* const* h_dw, int32_t pooled_tc, void* stream);
/* The same two passes with their two launches each fused into ONE (F1 -> F2, B1 -> B2): the workgroups of a clip meet on a
 * device-scope counter inside the launch (the second half's staging -- and in the backward the whole G branch -- runs while
 * the first half's tiles finish).  d_sync: >= 2 * N uint32, ZERO when first used (the kernels leave it zero), not shared by
 * launches that may run concurrently (one per stream).  Results are bit-identical to the unfused entry points.
 * VITTA_ERR_UNSUPPORTED when the launch could not be resident at once (N * tiles > 512 or LDS > 160 KiB). */
/* 1 if the fused forward AND backward launches below can hold every workgroup of N clips resident (they meet on a device
 * counter per clip): grid <= half of (occupancy query per CU x CUs), the other half being left to a second fused launch on
 * another stream.  0: use the two-launch forms (vitta_tam_branch_fwd_f32 / _bwd_f32); the fused entry points return
 * VITTA_ERR_UNSUPPORTED themselves when asked anyway. */
int vitta_tam_branch_fused_supported(int32_t N, int32_t C, int32_t T);
/* The L branch's weight gradients as their own launch (the backward entry points called with NULL dw0 / dw3 leave them out):
 * d_dw0 [C/4, C, 3] += sum over clips and t of dpre[n, o, t] pooled[n, c, t + j - 1];  d_dw3 [C, C/4] += sum of
 * (d gate * gate * (1 - gate))[n, c, t] h[n, o, t].  d_hact = the SECOND half of the forward's d_hpre (after N * C/4 * T floats),
 * d_dpre = the scratch behind d_gpooled's first N * C * T floats that the backward left.  Single writer per element (plain
 * read-modify-write): may run on another stream than the backward, after it. */
int vitta_tam_branch_wgrad_f32(const float* d_pooled, int32_t pooled_tc, const float* d_gate, const float* d_ggate, const float* d_hact,
                               const float* d_dpre, int32_t N, int32_t C, int32_t T, float* d_dw0, float* d_dw3, void* stream);
int vitta_tam_branch_fwd_fused_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                                   const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                                   const float* d_w3, int32_t N, int32_t C, int32_t T, float* d_kern, float* d_gate,
                                   float* d_hpre, void* d_sync, int32_t pooled_tc, void* stream);
int vitta_tam_branch_bwd_fused_f32(const float* d_pooled, const float* d_wg1, const float* const* h_bn_g, float eps_g,
                                   const float* d_wg3, const float* d_w0, const float* const* h_bn_l, float eps_l,
                                   const float* d_w3, int32_t N, int32_t C, int32_t T, const float* d_kern,
                                   const float* d_gate, const float* d_hpre, const float* d_gkern, const float* d_ggate,
                                   float* d_gpooled, float* const* h_dbn, float* const* h_dw, void* d_sync, int32_t pooled_tc, void* stream);


/* --------------------------------------------------------------------------
 * A8 glue -- eval-mode BatchNorm2d fused with its neighbours and with the ViTTA statistics.
 * During adaptation every BatchNorm is in eval() (corpus/basics.py:606-611).  One pass replaces
 * net.bnK(x) [+ identity] [-> relu] of TemporalBottleneck.forward (temporal_module.py:88-104) and, for a
 * hooked layer, the moments pass of A1 (the BN output y is consumed from registers, never stored);
 * the backward pass replaces ReLU-backward + the A6 injection + BatchNorm-backward.
 *   x [outer, C, HW] conv output; res [outer, C, HW] or NULL; z = act(bn(x) + res)
 *   d_triples: NULL, or the layer's (n, mean, M2) partial area of a plan workspace
 *              (float* workspace + 3*ws_off, geometry from vitta_plan_layer_geometry; nsplit must match)
 *   backward: d_gres NULL iff no residual; d_mu/d_coef_a/d_coef_b/d_gscale NULL iff not hooked;
 *             d_gx NULL: the BN input needs no gradient (first layer behind a frozen conv): only d gamma / d beta;
 *             d_gz2: NULL, or a second gradient of z (z feeds the next block's conv AND its identity path; summing
 *             the two here saves autograd's add pass over the largest tensors of the network);
 *             d_z needed only for relu+residual; d_partial: vitta_bn_act_partial_floats() floats of scratch;
 *             accumulate != 0: d_dgamma / d_dbeta += with fp32 atomics from the one launch (the caller hands the
 *             parameters' live .grad storage -- what autograd's AccumulateGrad would do with one more launch per
 *             tensor; d_partial may be NULL); accumulate == 0: partials + a deterministic fp64 finalize launch.
 * Requires C*HW % 4 == 0 (VITTA_ERR_UNSUPPORTED otherwise: use the unfused ops).
 * -------------------------------------------------------------------------- */
size_t vitta_bn_act_partial_floats(int64_t outer, int32_t C, int64_t HW, int32_t nsplit);
int vitta_bn_act_fwd_f32(const float* d_x, const float* d_res, float* d_z, const float* d_weight, const float* d_bias,
                         const float* d_rmean, const float* d_rvar, float eps, int64_t outer, int32_t C, int64_t HW,
                         int32_t nsplit, int32_t relu, float* d_triples, void* stream);
int vitta_bn_act_bwd_f32(const float* d_x, const float* d_z, const float* d_gz, const float* d_gz2, float* d_gx, float* d_gres,
                         const float* d_weight, const float* d_bias, const float* d_rmean, const float* d_rvar,
                         float eps, const float* d_mu, const float* d_coef_a, const float* d_coef_b,
                         const float* d_gscale, int64_t outer, int32_t C, int64_t HW, int32_t nsplit, int32_t relu,
                         float* d_partial, float* d_dgamma, float* d_dbeta, int32_t accumulate, void* stream);
/* out5 = {nsplit, nchunks, slots, ws_off (in triples), vec} of one layer of a plan */
int vitta_plan_layer_geometry(const vitta_plan* plan, int layer, int64_t* out5);

/* Same attention with the additive terms kept ON CHIP: the relative-position bias is
 * table[code[q] - code[k] + code_off][h] (the index of swin_transformer.py:113-124 is linear in the
 * token coordinates; code[t] = (t_d*(2wh-1) + t_h)*(2ww-1) + t_w) and the shift mask is -100 where
 * region[b % nW][q] != region[b % nW][k] (swin_transformer.py:316-329).  No [N x N] operand exists.
 *   d_table [T, nH] (the module's parameter as stored), T <= 4096; d_code int32 [N];
 *   d_region int32 [nW, N] or NULL; d_dtable [T, nH] or NULL (accumulated: zero it first).
 *   d_rowmap int32 [map_windows, N] or NULL.  NULL: window b, token n is row b*N + n of qkv / out / dqkv (the
 *   partitioned layout of swin_transformer.py:233).  Otherwise it is row (b / map_windows)*tokens_per_sample +
 *   rowmap[b % map_windows][n] of the natural [B, D*H*W] token order: torch.roll + window_partition and their
 *   inverses (swin_transformer.py:222-243) become address arithmetic (qkv / proj are per token and commute with
 *   the permutation); tokens_per_sample == map_windows * N (no padding). */
/* the relative-position form also covers windows of 401..800 tokens (e.g. (16,7,7) = 784: keys / queries walked in
 * chunks through LDS, online softmax), table T <= 8192; the dense form and T stay at N <= 400, T <= 4096 otherwise */
int vitta_wmsa_rel_supported(int32_t N, int32_t head_dim);
int vitta_wmsa_rel_fwd_f32(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code,
                           int32_t code_off, const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH,
                           int32_t head_dim, float scale, const int32_t* d_rowmap, int32_t map_windows,
                           int64_t tokens_per_sample, float* d_out, float* d_lse, void* stream);
int vitta_wmsa_rel_bwd_f32(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code,
                           int32_t code_off, const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH,
                           int32_t head_dim, float scale, const int32_t* d_rowmap, int32_t map_windows,
                           int64_t tokens_per_sample, const float* d_out, const float* d_dout, const float* d_lse,
                           float* d_delta, float* d_dqkv, float* d_dtable, void* stream);

/* bfloat16-OPERAND form of the two entry points above (BASELINE config 5: window (16,7,7) = 784 tokens): q, k, v, dO are
 * rounded to bf16 while they are staged, both GEMMs of every direction run on v_mfma_f32_16x16x16_bf16 with fp32
 * accumulation; softmax, bias / mask terms, lse, delta and all outputs are fp32.  Windows up to 800 tokens in one pass (K
 * and V fit LDS together at 2 bytes per element).  The bias table is an input only (no d_dtable: train it with the fp32
 * entry points).  vitta_wmsa_bf16_supported: head_dim == 32, N <= 800, table_rows <= 8192 and the three kernels' LDS
 * carves within 160 KiB. */
int vitta_wmsa_bf16_supported(int32_t N, int32_t head_dim, int32_t table_rows);
int vitta_wmsa_rel_fwd_bf16(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                            const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                            float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                            float* d_out, float* d_lse, void* stream);
int vitta_wmsa_rel_bwd_bf16(const float* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                            const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                            float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                            const float* d_out, const float* d_dout, const float* d_lse, float* d_delta, float* d_dqkv,
                            void* stream);
/* The same kernels with 2-byte activations on both sides (io_bf16 != 0: d_qkv, d_out, d_dout, d_dqkv are bfloat16 tensors of the same
 * shapes; lse / delta stay fp32) -- the attention of the bf16 data flow sits between two dense products that hand over bfloat16. */
int vitta_wmsa_rel_fwd_bf16_io(const void* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                               const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                               float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                               void* d_out, float* d_lse, int32_t io_bf16, void* stream);
int vitta_wmsa_rel_bwd_bf16_io(const void* d_qkv, const float* d_table, int32_t T, const int32_t* d_code, int32_t code_off,
                               const int32_t* d_region, int32_t nW, int64_t B_, int32_t N, int32_t nH, int32_t head_dim,
                               float scale, const int32_t* d_rowmap, int32_t map_windows, int64_t tokens_per_sample,
                               const void* d_out, const void* d_dout, const float* d_lse, float* d_delta, void* d_dqkv,
                               float* d_dtable, float* d_dtable_ws, int64_t dtable_ws_bytes, int32_t io_bf16, void* stream);
/* d_dtable [T, nH] fp32 or NULL: the gradient of the relative-position table is ADDED to it (swin_transformer.py:110-151 under
 * SGD over all parameters) -- by the one-pass backward only: d bias = dS summed along the score tile's diagonals in registers, binned
 * in LDS per (window, head); VITTA_ERR_UNSUPPORTED where that form does not apply (vitta_wmsa_bf16_dtable_supported says so
 * beforehand).  d_dtable_ws: caller-owned scratch of vitta_wmsa_bf16_dtable_workspace_bytes(B_, nH, T) bytes (contents undefined before
 * and after) -- the pairs' columns leave as plain stores and one reduce launch adds the windows; NULL / too small: one global atomic per
 * (pair, entry), ~6x slower at 256 windows per head. */
size_t vitta_wmsa_bf16_dtable_workspace_bytes(int64_t B_, int32_t nH, int32_t table_rows);
int vitta_wmsa_bf16_dtable_supported(int32_t N, int32_t head_dim, int32_t table_rows);

/* --------------------------------------------------------------------------
 * A10, SGD over all parameters on the bf16 recipe -- dense WEIGHT gradients with 2-byte operands (round 6).
 * Replaces autograd's d weight (and d bias) of the nn.Linear products of swin_transformer.py:30-35, 144, 165, 304-311 where both the
 * layer input x [M, K] and the output gradient g [M, N] are bfloat16 in memory (ops.bf16_flow): out [N, K] fp32 (+)= g^T x on
 * v_mfma_f32_16x16x16_bf16, both fragments by the LDS transpose read (the contraction runs over the tokens, the slow axis of both
 * operands), fp32 accumulation; d_bias_grad [N] fp32 or NULL: += column sums of g.  M % 32 == 0, N % 128 == 0, K % 128 == 0.
 * The token axis is cut into splits; d_ws: caller-owned scratch of vitta_gemm_tn_bf16_workspace_bytes(M, N, K) bytes (16-byte aligned,
 * contents undefined before and after) for the splits' partial tiles, which one reduce launch adds in split order.
 * -------------------------------------------------------------------------- */
int vitta_gemm_tn_bf16_supported(int64_t M, int32_t N, int32_t K);
size_t vitta_gemm_tn_bf16_workspace_bytes(int64_t M, int32_t N, int32_t K);
int vitta_gemm_tn_bf16(const void* d_g, const void* d_x, float* d_out, int64_t M, int32_t N, int32_t K, int32_t accumulate,
                       float* d_bias_grad, void* d_ws, size_t ws_bytes, void* stream);

/* --------------------------------------------------------------------------
 * A7 -- optimizer update on the flat parameter arena, one launch.
 * Replaces optimizer.step() of corpus/basics.py:671 for the optimizers built at corpus/basics.py:547-560 (torch.optim.Adam over the affine tensors /
 * torch.optim.SGD over every parameter), same element-wise arithmetic as torch's single-tensor formulation.
 *   vitta_adam_step_f32: t = d_step[0] + 1 (device scalar, incremented by the call, so a captured graph advances it: ONE
 *                        float); d_ticket: ONE uint32 of the caller's, zero when first used and left zero -- the launch's arrival
 *                        counter: the last workgroup to arrive writes the new step (one launch; a separate word so that a 1-float step
 *                        tensor -- a loaded or replaced optimizer state -- is never written out of bounds; NULL = invalid argument);
 *                        g' = g + wd p; m = lerp(m, g', 1-b1); v = b2 v + (1-b2) g'^2;
 *                        p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  m, v, *d_step start at 0.
 *   vitta_sgd_step_f32:  d = g + wd p; buf = momentum buf + d (buf starts at 0); p -= lr buf.
 *                        momentum == 0: d_momentum_buf may be NULL.
 * All arrays n floats, 16-byte aligned.
 * -------------------------------------------------------------------------- */
int vitta_adam_step_f32(float* d_param, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, float* d_step,
                        uint32_t* d_ticket, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t n, void* stream);
int vitta_sgd_step_f32(float* d_param, const float* d_grad, float* d_momentum_buf, float lr, float momentum,
                       float weight_decay, int64_t n, void* stream);

/* --------------------------------------------------------------------------
 * A10 -- residual update with per-sample stochastic depth (Video Swin blocks, swin_transformer.py:268-275 with timm's
 * DropPath): out[b] = x[b] + scale[b] * branch[b], one pass.  d_scale [samples] device array or NULL (= 1);
 * d_x NULL: out[b] = scale[b] * branch[b] (the backward of the branch).  per_sample % 4 == 0, 16-byte aligned.
 * -------------------------------------------------------------------------- */
int vitta_scale_add_f32(const float* d_x, const float* d_branch, const float* d_scale, int64_t samples,
                        int64_t per_sample, float* d_out, void* stream);
/* PatchMerging's gather (swin_transformer.py:281-286: cat of x[:, :, 0::2, 0::2], [1::2, 0::2], [0::2, 1::2], [1::2, 1::2] along the
 * channels) on channels-last planes: merged[p][i][j][k C + c] = x[p][2 i + (k & 1)][2 j + (k >> 1)][c]; x [planes][2 H2][2 W2][C],
 * merged [planes][H2][W2][4 C].  inverse = 0: d_src = x, d_dst = merged;  1: d_src = merged, d_dst = x (the gradient).  One launch
 * (torch: four strided copies each way).  C % 4 == 0, 16-byte aligned. */
int vitta_patch_gather_f32(const float* d_src, float* d_dst, int64_t planes, int32_t H2, int32_t W2, int32_t C, int32_t inverse,
                           void* stream);

/* --------------------------------------------------------------------------
 * A8 -- the 2D convolutions of the TANet trunk (torchvision ResNet-50 Bottleneck under models/tanet_models/tanet.py:125-150,
 * wrapped by TemporalBottleneck, temporal_module.py:85-106: conv1x1 -> BN -> ReLU -> TAM -> conv3x3 -> BN -> ReLU ->
 * conv1x1 -> BN -> (+identity) -> ReLU) as ONE implicit-GEMM kernel family on v_mfma_f32_32x32x2_f32 (exact fp32).
 *
 * Activations are "channel-major planes": tensor[c][p], p = frame * H*W + h * W + w over ALL frames of the clip, so a
 * pointwise convolution is one GEMM  D[p][k] = sum_c X[c][p] W[c][k]  with pixels on the MFMA row axis (every epilogue
 * access is 4 consecutive pixels of one channel per lane = 16 bytes) and a 3x3 / strided / transposed convolution the
 * same GEMM with a gathered A operand (tap table below).  Weights are PACKED [tap][C][K] (K contiguous):
 *   forward 1x1 : W^T of the [K, C, 1, 1] parameter;  dgrad 1x1: the [K, C] parameter itself (roles of C and K swap);
 *   forward 3x3 : w[k][c][dh][dw] -> [dh*3+dw][c][k];  dgrad 3x3: [tap'][k][c] = w[k][c][2-dh'][2-dw'].
 *
 * The M axis walks a grid of (n, i, j), i < Hg, j < Wg, per frame.  Source pixel of tap t: (i*sstride + dh[t],
 * j*sstride + dw[t]) in the Hs x Ws planes of x (zero outside); destination pixel (i*ostride + oa, j*ostride + ob) in
 * the Hy x Wy planes of y.  ostride 2 (one launch per parity class) is the data gradient of a stride-2 convolution.
 *
 * Epilogue (flags):
 *   forward : raw = acc                                   -> y_raw (optional second output)
 *             z = raw * s_k + t_k  (epi_bn, eval-mode BatchNorm2d: s = gamma / sqrt(var + eps), t = beta - mean * s)
 *             VITTA_CONV_STATS: st_s1[k] += sum(z - shift_k), st_s2[k] += sum (z - shift_k)^2 over the tile's pixels
 *                               (the hooked-layer moments of utils/norm_stats_utils.py:185-253, additive form);
 *                               with VITTA_CONV_STATS_RAW the same sums of raw - shift_k: moments of the BatchNorm INPUT
 *                               (before_norm hooks, utils/norm_stats_utils.py:52-53), taken directly -- no division by gamma
 *             o = (VITTA_CONV_EPI_APPLY ? z : raw) (+ res) ; VITTA_CONV_EPI_RELU: max(o, 0)          -> y
 *   backward (VITTA_CONV_BWD_BN; acc = gradient w.r.t. the ACTIVATED input a = relu(bn(bwd_x)) of the forward conv):
 *             g = acc (+ res: the gradient arriving over the identity path)
 *             z = bwd_x * s + t ; m = relu mask (z > 0, or bwd_mask > 0 when given; 1 without VITTA_CONV_BWD_RELU)
 *             dz = g * m + gscale * (a_k + b_k (z - mu_k))   (statistics-loss gradient, A6; inj_* NULL: none)
 *             dgamma_k += sum dz * (bwd_x - mean_k) * rstd_k ; dbeta_k += sum dz   (atomics, may point into .grad)
 *             y = dz * s_k ; y_raw (optional) = g * m
 *   VITTA_CONV_PRO_BN_RELU: x is the RAW output of the previous convolution; relu(bn(x)) (pro_bn) is applied on load.
 * All pointers are device pointers; bn arrays are {gamma, beta, running_mean, running_var}.
 * -------------------------------------------------------------------------- */
#define VITTA_CONV_PRO_BN_RELU 1
#define VITTA_CONV_EPI_APPLY 2
#define VITTA_CONV_EPI_RELU 4
#define VITTA_CONV_STATS 8
#define VITTA_CONV_RES 16
#define VITTA_CONV_RES_HALF 32 /* res is a half-resolution tensor [K][N * ceil(Hy/2) * ceil(Wy/2)] added at even (h, w) */
#define VITTA_CONV_BWD_BN 64
#define VITTA_CONV_BWD_RELU 128
/* The four parity classes of a stride-2 data gradient in ONE launch (conv_b3.hip only; vitta_conv_supported says whether
 * the descriptor qualifies): ostride must be 2, the taps are listed class by class -- class c = 2 a + b writes output
 * pixels (2 i + a, 2 j + b) and owns cls_ntaps[c] >= 1 consecutive entries of the tap table (oa / ob are ignored). */
#define VITTA_CONV_PARITY4 256
#define VITTA_CONV_STATS_RAW 512 /* with VITTA_CONV_STATS: the sums are of the raw convolution output, not of z */
/* With VITTA_CONV_BWD_BN and inj_*: the hooked feature is the RAW input of the BatchNorm (before_norm hooks,
 * utils/norm_stats_utils.py:185): gscale (a_k + b_k (x - mu_k)) is added to the output gradient itself, d gamma / d beta do not see it. */
#define VITTA_CONV_INJ_RAW 1024
/* Forward launches with a contiguous output and epi_bn: the per-(frame, channel) means of relu(z), z = the eval-mode BatchNorm of
 * the raw output -- TAM's adaptive average pooling of relu(bn1(conv1(x))) (temporal_module.py:53: F.adaptive_avg_pool2d) taken from
 * the accumulators instead of a pass over x1 -- ADDED (atomics) to pool[N frames][K] of int64 FIXED-POINT numbers with 32
 * fractional bits (order-independent sums: the means are forward activations), frame-major (the 32 channels a half-wave adds are
 * contiguous; channel-major costs 10-18x, tools/ubench/atomic_line_probe.hip), which the caller zeroes; pool_scale = 1 / (Hy * Wy).
 * Hy * Wy >= 32.  The TAM branch entry points read this tensor with pooled_tc = 1. */
#define VITTA_CONV_POOL 2048
#define VITTA_CONV_MAX_TAPS 9

typedef struct vitta_conv_desc {
  const float* x;      /* [C][x_pixels]   x_pixels = N * Hs * Ws */
  const float* w;      /* packed [n_wtaps][C][K] */
  float* y;            /* [K][y_pixels]   y_pixels = N * Hy * Wy */
  float* y_raw;        /* optional second output, same shape as y */
  const float* res;    /* optional addend, same shape as y (or half resolution, VITTA_CONV_RES_HALF) */
  const float* pro_bn[4];
  const float* epi_bn[4];
  const float* bwd_bn[4];
  float pro_eps, epi_eps, bwd_eps;
  const float* st_shift; /* [K] */
  float* st_s1;          /* [K] */
  float* st_s2;          /* [K] */
  const float* bwd_x;    /* [K][y_pixels] raw convolution output whose BatchNorm is differentiated */
  const float* bwd_mask; /* [K][y_pixels] or NULL */
  const float* inj_mu;   /* [K] or NULL */
  const float* inj_a;
  const float* inj_b;
  const float* inj_gscale; /* device scalar */
  float* dgamma;         /* [K] accumulated */
  float* dbeta;
  int32_t C, K, N;
  int32_t Hs, Ws, Hg, Wg, Hy, Wy;
  int32_t sstride, ostride, oa, ob;
  int32_t ntaps;
  int8_t dh[VITTA_CONV_MAX_TAPS], dw[VITTA_CONV_MAX_TAPS], wt[VITTA_CONV_MAX_TAPS]; /* wt: tap slot in w */
  int32_t flags;
  int32_t tile; /* 0: library's choice; else (BM << 16) | BN */
  /* Split-K: with few output tiles (the 14x14 / 7x7 stages) several workgroups share a tile, each walking a slice of K;
   * partial tiles meet in the last-arriving workgroup through `workspace` (arrival counters + slabs).  The workspace
   * is caller-owned, must be ZERO when first used (its first 64 KiB hold the counters, which return to zero after every
   * launch; slabs follow), >= 256-byte aligned,
   * and must not be shared by launches that may run concurrently (one per stream).  NULL / too small: no split. */
  int32_t ksplit; /* 0: library's choice; 1: never split; n: exactly n slices (VITTA_ERR_WORKSPACE if it does not fit);
                   * -1: never split, and persistent workgroups where the split-bf16 pointwise form has them (tools / tests) */
  void* workspace;
  int64_t workspace_bytes;
  /* Optional split-bf16 image of the same weights (vitta_conv_pack_b3 of the packed [n_wtaps][C][K] array).  When given
   * and the shape qualifies (C % 32 == 0, K % 64 == 0, no VITTA_CONV_PRO_BN_RELU, tile 0 or 128 x 64) the launch runs on
   * the bf16 matrix pipe with every fp32 operand split into three bf16 terms and six products per multiply-add (fp32
   * accumulation; error of the fp32-roundoff class, see conv_b3.hip); `w` is then not read.  NULL: exact fp32 MFMA. */
  const void* w_b3;
  int8_t cls_ntaps[4]; /* VITTA_CONV_PARITY4: taps of class 0..3 (sum = ntaps) */
  void* pool;        /* VITTA_CONV_POOL: int64 [N][K], 32 fractional bits, accumulated */
  float pool_scale;  /* 1 / (Hy * Wy) */
} vitta_conv_desc;

/* 1 if the shape is covered: C % 16 == 0 (or C < 16 handled by the stem entry), K % 32 == 0, pixel counts % 4 == 0. */
int vitta_conv_supported(const vitta_conv_desc* h_desc);
int vitta_conv_f32(const vitta_conv_desc* h_desc, void* stream);
/* The same launch with two caller-created events (vitta_event_create) attached to the kernel's own dispatch: their elapsed
 * time is the kernel's duration as rocprofv3 --kernel-trace reports it (bench.py's live roofline figure). */
int vitta_conv_timed_f32(const vitta_conv_desc* h_desc, void* stream, void* ev_start, void* ev_stop);
/* Multiply-add count x 2 of the launch (algorithmic: N * Hg * Wg output positions x K x C x ntaps; padding taps included,
 * tile tails excluded). */
int64_t vitta_conv_flops(const vitta_conv_desc* h_desc);
/* Workspace the library's split choice (or h_desc->ksplit) needs for this descriptor; 0 = none. */
size_t vitta_conv_workspace_bytes(const vitta_conv_desc* h_desc);
/* Which kernel family the launch of this descriptor runs on (tests assert the path under test; -1: unsupported). */
#define VITTA_CONV_KERNEL_TILE 0 /* conv.hip: exact fp32 MFMA, tile per workgroup (+ split K) */
#define VITTA_CONV_KERNEL_SK 1   /* conv_sk.hip: exact fp32 MFMA, persistent stream-K */
#define VITTA_CONV_KERNEL_PW 2   /* conv_pw.hip: exact fp32 MFMA, pointwise tile per workgroup */
#define VITTA_CONV_KERNEL_B3 3   /* conv_b3.hip: split-bf16 operands on the bf16 matrix pipe */
int vitta_conv_kernel(const vitta_conv_desc* h_desc);
/* Host only: n / d as the convolution kernels compute it on the device (multiply-high by the host-made reciprocal of a launch
 * constant, conv_common.h FastDiv), for 0 <= n < 2^31, 1 <= d < 2^31; -1 outside that range.  The CPU suite holds it to n // d. */
int64_t vitta_conv_fastdiv_host(int64_t n, int64_t d);
/* Workgroups the launch of this descriptor would use (for tile selection / tests). */
int64_t vitta_conv_num_blocks(const vitta_conv_desc* h_desc);

/* --------------------------------------------------------------------------
 * A8 -- weight gradients of the same convolutions (the reference's default optimizer trains every parameter,
 * corpus/basics.py:547-560):   grad_w[k][c][wt[t]] += sum_p A[c][src(p, t)] * dy[k][p]
 * over the N * Hg * Wg output positions p of the forward convolution; A = x, or relu(bn(x)) with VITTA_CONV_PRO_BN_RELU
 * (pro_bn as in vitta_conv_desc).  x [C][N * Hs * Ws] and dy [K][N * Hg * Wg] are channel-major planes; grad_w is the
 * parameter's own [K][C][wtaps] layout (wtaps = kh * kw) and is ACCUMULATED with atomics (may point into .grad).
 * Tap t reads source pixel (i * sstride + dh[t], j * sstride + dw[t]) of output position (i, j).  Unless the convolution
 * is pointwise with stride 1, the caller supplies two per-position tables (device, int32 [N * Hg * Wg]):
 *   src_off[p]  = n * Hs * Ws + (i * sstride) * Ws + j * sstride        (source pixel of tap (0, 0))
 *   src_mask[p] = bit t set when tap t's source pixel lies inside the plane.
 * C % 64 == 0, K % 64 == 0, pixel counts % 4 == 0.
 * -------------------------------------------------------------------------- */
typedef struct vitta_wgrad_desc {
  const float* x;
  const float* dy;
  float* grad_w;
  const float* pro_bn[4];
  float pro_eps;
  const int32_t* src_off;
  const int32_t* src_mask;
  int32_t C, K, N;
  int32_t Hs, Ws, Hg, Wg;
  int32_t sstride;
  int32_t ntaps, wtaps;
  int8_t dh[VITTA_CONV_MAX_TAPS], dw[VITTA_CONV_MAX_TAPS], wt[VITTA_CONV_MAX_TAPS];
  int32_t flags; /* VITTA_CONV_PRO_BN_RELU, VITTA_WGRAD_DEFER_REDUCE or 0 */
  /* optional scratch (>= 768 x 32 KiB = 24 MiB on an MI355X; no initial content required, one per stream): the workgroups'
   * partial tiles meet there and a second launch adds them to grad_w in a fixed order; without it every partial tile is
   * added to grad_w with atomics (thousands of adds per weight on the small early layers) */
  void* workspace;
  int64_t workspace_bytes;
} vitta_wgrad_desc;
int vitta_conv_wgrad_f32(const vitta_wgrad_desc* h_desc, void* stream);
/* VITTA_WGRAD_DEFER_REDUCE in flags: vitta_conv_wgrad_f32 leaves its partial tiles in `workspace` and skips the second launch;
 * vitta_conv_wgrad_reduce_f32 then adds the partial tiles of up to four such launches (the SAME descriptors, each with a
 * workspace of its own, all issued earlier on this stream) to their grad_w in ONE launch -- the weight gradients of one
 * bottleneck are four launches + one instead of four + four. */
#define VITTA_WGRAD_DEFER_REDUCE 1024
int vitta_conv_wgrad_reduce_f32(const vitta_wgrad_desc* const* h_descs, int32_t n, void* stream);

/* Split-bf16 weight image for vitta_conv_desc::w_b3.  d_src: fp32 [taps][R][O] (R = reduction channels, O = output
 * channels: the packed forward / data-gradient arrays of vitta_conv_desc::w), R % 32 == 0.
 * d_dst: [taps][R / 32][3 planes hi | mid | lo][4 channel octets][O][8] bfloat16 = vitta_conv_pack_b3_bytes() bytes,
 * x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even).
 * _table: many weights in one launch (trainable weights are re-split every step); entries ordered by first_unit =
 * number of 16-byte units (taps * R / 8 * O each) of all entries before this one. */
typedef struct vitta_pack_b3_entry {
  const float* src;
  void* dst;
  int64_t first_unit;
  int32_t taps, R, O, pad;
} vitta_pack_b3_entry;
size_t vitta_conv_pack_b3_bytes(int32_t taps, int32_t R, int32_t O);
int vitta_conv_pack_b3(const float* d_src, void* d_dst, int32_t taps, int32_t R, int32_t O, void* stream);
int vitta_conv_pack_b3_table(const vitta_pack_b3_entry* d_table, int32_t n_entries, int64_t total_units, void* stream);

/* Packed copies of MANY convolution weights in one launch (trainable weights are re-packed every step).  d_table: device
 * array of entries; `first` = number of weight elements of all entries before this one (entries ordered by it);
 * dst_fwd [taps][C][K], dst_bwd [taps][K][C] or NULL (a 1x1 weight is its own backward pack); src the [K][C][taps]
 * parameter.  total_elements = sum of K * C * taps. */
typedef struct vitta_repack_entry {
  const float* src;
  float* dst_fwd;
  float* dst_bwd;
  int64_t first;
  int32_t K, C, taps, pad;
} vitta_repack_entry;
int vitta_conv_repack_f32(const vitta_repack_entry* d_table, int32_t n_entries, int64_t total_elements, void* stream);

/* --------------------------------------------------------------------------
 * A8 / A9 on channel-major planes (the layout of vitta_conv_f32: tensor[c][f * HW + hw], f = n * T + t).
 * TemporalBottleneck (temporal_module.py:85-106) with conv1 writing its RAW output x1: the TAM kernels apply
 * a = relu(bn1(x1)) while loading (h_bn = {gamma, beta, running_mean, running_var} device pointers [C]):
 *   vitta_tam_pool_cm     d_pool [N, C, T] = mean_hw a                                   (temporal_module.py:47-52)
 *   vitta_tam_agg_fwd_cm  d_out[c][n,t,:] = sum_j K[n,c,j] gate[n,c,t+j-1] a[c][n,t+j-1,:]   (:56-63)
 *   vitta_tam_agg_bwd_cm  d_ga = d a (without the pooling path), d_ggate [N,C,T] (+ N*C*T*3 floats of scratch behind it),
 *                         d_gkern [N*C, 3]
 * vitta_bn_bwd_cm: backward of y = [relu](bn_eval(x)) in this layout, the element-wise piece between two data-gradient
 * convolutions:  g = d_g (+ d_g2) (+ rowadd_scale * d_rowadd[n, c, t], the TAM pooling gradient with scale 1 / HW);
 *   m = relu ? (d_mask ? d_mask > 0 : z > 0) : 1 ;  dz = g m + gscale (a_c + b_c (z - mu_c)) (statistics-loss gradient, A6);
 *   d_dgamma[c] += sum dz x_hat ; d_dbeta[c] += sum dz (atomics) ; d_dx = dz s_c ; d_gm (optional) = g m.
 *   relu: bit 0 = the ReLU mask; bit 1 (VITTA_BN_BWD_INJ_RAW) = the hooked feature is the raw input x (before_norm hooks,
 *   utils/norm_stats_utils.py:185): dz = g m, d_dx = dz s_c + gscale (a_c + b_c (x - mu_c)).
 * vitta_avgpool_cm(_bwd): the trunk's AdaptiveAvgPool2d(1) (tanet.py:147) from planes [C][F*HW] to features [F, C].
 * -------------------------------------------------------------------------- */
#define VITTA_BN_BWD_INJ_RAW 2
int vitta_tam_pool_cm_f32(const float* d_x, const float* const* h_bn, float eps, int32_t C, int32_t N, int32_t T, int32_t HW,
                          float* d_pool, void* stream);
int vitta_tam_agg_fwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, const float* d_gate, const float* d_kern,
                             int32_t C, int32_t N, int32_t T, int32_t HW, float* d_out, void* stream);
int vitta_tam_agg_bwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, const float* d_gate, const float* d_kern,
                             const float* d_gout, int32_t C, int32_t N, int32_t T, int32_t HW, float* d_ga, float* d_ggate,
                             float* d_gkern, void* stream);
int vitta_bn_bwd_cm_f32(const float* d_g, const float* d_g2, const float* d_x, const float* d_mask, const float* d_rowadd,
                        float rowadd_scale, const float* const* h_bn, float eps, const float* d_mu, const float* d_coef_a,
                        const float* d_coef_b, const float* d_gscale, int32_t relu, float* d_dx, float* d_gm, float* d_dgamma,
                        float* d_dbeta, int32_t C, int32_t N, int32_t T, int32_t HW, void* stream);
int vitta_avgpool_cm_f32(const float* d_x, int32_t C, int32_t F, int32_t HW, float* d_feat, void* stream);
int vitta_avgpool_cm_bwd_f32(const float* d_gfeat, int32_t C, int32_t F, int32_t HW, float* d_gx, void* stream);

/* --------------------------------------------------------------------------
 * A2 / A10 -- LayerNorm over the channel axis of channels-last rows [rows, C], fused with its surroundings in a Video
 * Swin block (swin_transformer.py:245-275) and with the ViTTA statistics of a hooked LayerNorm
 * (utils/norm_stats_utils.py:222-230).  C in {128, 256, 512, 1024, 2048}.
 *   forward : x' = x + scale[row / rows_per_sample] * branch   (d_branch/d_xnew NULL: x' = x, nothing written)
 *             y = LN(x') * gamma + beta ; d_mean / d_rstd [rows] saved for the backward;
 *             hooked: d_shift [C] (source mean) and d_partial [vitta_ln_num_partials(rows)][2][C] receive per-workgroup
 *             sums of (y - shift), (y - shift)^2; vitta_colsum2_f32 adds them into the s1 / s2 statistics.
 *   backward: g = g_y (+ gscale (a_c + b_c (y - mu_c)) when d_mu != NULL);  d_gx = LN-backward(g) (+ d_gxnew);
 *             d_gbranch (optional) = scale * d_gx;  d_partial [..][2][C] = per-workgroup sums of g*xhat | g
 *             (-> vitta_colsum2_f32 -> d gamma, d beta).
 * vitta_colsum2_f32: out_a[c] += sum_b partial[b][0][c], out_b[c] += sum_b partial[b][1][c] (row-parallel, one fp32
 *             atomic per column and 32 partial rows: zero the outputs for a fresh sum, or hand live gradient storage);
 *             d_cnt (optional) <- cnt_value (the plan's per-layer sample count).
 * -------------------------------------------------------------------------- */
int vitta_ln_supported(int32_t C);
int64_t vitta_ln_num_partials(int64_t rows);
int vitta_ln_fwd_f32(const float* d_x, const float* d_branch, const float* d_scale, int64_t rows, int64_t rows_per_sample,
                     int32_t C, const float* d_gamma, const float* d_beta, float eps, float* d_xnew, float* d_y,
                     float* d_mean, float* d_rstd, const float* d_shift, float* d_partial, void* stream);
int vitta_ln_bwd_f32(const float* d_gy, const float* d_gxnew, const float* d_x, const float* d_mean, const float* d_rstd,
                     const float* d_gamma, const float* d_beta, const float* d_scale, const float* d_mu,
                     const float* d_coef_a, const float* d_coef_b, const float* d_gscale, int64_t rows,
                     int64_t rows_per_sample, int32_t C, float* d_gx, float* d_gbranch, float* d_partial, void* stream);
/* The same passes with bfloat16 on the sides that touch a dense product (the bf16 recipe of BASELINE config 5): the forward reads a
 * bf16 branch (VITTA_LN_BRANCH_BF16) and / or writes y as bf16 (VITTA_LN_Y_BF16: the LayerNorm output only feeds qkv / fc1); the
 * backward reads g_y as bf16 (VITTA_LN_GY_BF16: what the product's data gradient wrote) and / or writes d_gbranch as bf16
 * (VITTA_LN_GBRANCH_BF16: the gradient of a bf16 branch -- then written even without a scale).  Statistics, the residual stream x,
 * d_gx and every reduction stay fp32. */
#define VITTA_LN_BRANCH_BF16 1
#define VITTA_LN_Y_BF16 2
#define VITTA_LN_GY_BF16 4
#define VITTA_LN_GBRANCH_BF16 8
int vitta_ln_fwd_mixed(const float* d_x, const void* d_branch, const float* d_scale, int64_t rows, int64_t rows_per_sample,
                       int32_t C, const float* d_gamma, const float* d_beta, float eps, float* d_xnew, void* d_y,
                       float* d_mean, float* d_rstd, const float* d_shift, float* d_partial, int32_t flags, void* stream);
int vitta_ln_bwd_mixed(const void* d_gy, const float* d_gxnew, const float* d_x, const float* d_mean, const float* d_rstd,
                       const float* d_gamma, const float* d_beta, const float* d_scale, const float* d_mu,
                       const float* d_coef_a, const float* d_coef_b, const float* d_gscale, int64_t rows,
                       int64_t rows_per_sample, int32_t C, float* d_gx, void* d_gbranch, float* d_partial, int32_t flags,
                       void* stream);
int vitta_colsum2_f32(const float* d_partial, int64_t n_partials, int32_t C, float* d_out_a, float* d_out_b, float* d_cnt,
                      float cnt_value, void* stream);
/* The same sums for n_items partial matrices (a HOST array of items, read during the call) in one launch per 32 items: the
 * column sums of a pass's LayerNorm sites, deferred to the point where their results are first read (vitta_amd/ops.py:
 * ColsumQueue).  Items may share outputs (the sums are atomic adds). */
typedef struct {
  const float* d_partial; /* [n_partials][2][C] */
  int64_t n_partials;
  int32_t C;
  float cnt_value;
  float* d_out_a; /* [C] += */
  float* d_out_b; /* [C] += */
  float* d_cnt;   /* optional: <- cnt_value */
} vitta_colsum_item;
int vitta_colsum2_multi_f32(const vitta_colsum_item* h_items, int32_t n_items, void* stream);

/* --------------------------------------------------------------------------
 * A8 -- ResNet stem tail in one pass: eval BatchNorm -> ReLU -> MaxPool2d(3, stride 2, pad 1) over the 7x7 convolution's
 * output (torchvision ResNet.forward: bn1, relu, maxpool).  d_x [N, C, H, W] -> d_out [N, C, PH, PW],
 * PH = (H - 1) / 2 + 1.  The backward produces ONLY d gamma / d beta (ACCUMULATED with atomics; affine-only adaptation:
 * the stem convolution is frozen, nothing flows below the BatchNorm), recomputing each window's maximum from d_x:
 * max-pool's scatter becomes a reduction.  N * C <= 65535.
 * -------------------------------------------------------------------------- */
int vitta_stem_bn_relu_pool_fwd_f32(const float* d_x, const float* const* h_bn, float eps, int64_t N, int32_t C, int32_t H,
                                    int32_t W, float* d_out, void* stream);
int vitta_stem_bn_relu_pool_bwd_affine_f32(const float* d_x, const float* d_gpool, const float* const* h_bn, float eps,
                                           int64_t N, int32_t C, int32_t H, int32_t W, float* d_dgamma, float* d_dbeta,
                                           void* stream);
/* The same backward that ALSO produces d_dy [N, C, H, W], the gradient w.r.t. the convolution output (trainable stem
 * convolution): ADDED into d_dy, which the caller zeroes (max-pool windows overlap).  d_dy NULL = the affine-only form.
 * W % 4 == 0, W <= 256. */
int vitta_stem_bn_relu_pool_bwd_f32(const float* d_x, const float* d_gpool, const float* const* h_bn, float eps, int64_t N,
                                    int32_t C, int32_t H, int32_t W, float* d_dgamma, float* d_dbeta, float* d_dy, void* stream);
/* The two passes with the POOLED tensor (forward output, backward upstream gradient) in channel-major planes
 * [C][N * PH * PW], the layout of the trunk's convolutions: no transposing copy between the stem and layer1.  Tiled path only
 * (W % 4 == 0, W <= 256, d_x 16-byte aligned: VITTA_ERR_UNSUPPORTED otherwise); d_dy as above (NULL: affine-only). */
int vitta_stem_bn_relu_pool_fwd_cm_f32(const float* d_x, const float* const* h_bn, float eps, int64_t N, int32_t C, int32_t H,
                                       int32_t W, float* d_out_cm, void* stream);
int vitta_stem_bn_relu_pool_bwd_cm_f32(const float* d_x, const float* d_gpool_cm, const float* const* h_bn, float eps, int64_t N,
                                       int32_t C, int32_t H, int32_t W, float* d_dgamma, float* d_dbeta, float* d_dy, void* stream);
/* Weight gradient of the stem convolution: d_dw [64, 3, 7, 7] += sum over frames and output pixels of d_dy [N, 64, OH, OW]
 * times the 7x7 / stride 2 / pad 3 patches of d_x [N, 3, H, W] (v_mfma_f32_32x32x2_f32; per-workgroup partial sums meet in
 * d_ws, >= vitta_stem_conv7_wgrad_workspace_bytes() bytes, no initial content required).  W % 4 == 0. */
size_t vitta_stem_conv7_wgrad_workspace_bytes(void);
int vitta_stem_conv7_wgrad_f32(const float* d_x, const float* d_dy, int64_t N, int32_t H, int32_t W, float* d_dw, void* d_ws,
                               size_t ws_bytes, void* stream);

/* --------------------------------------------------------------------------
 * A8 -- the stem convolution itself: Conv2d(3, 64, kernel 7, stride 2, pad 3, no bias), torchvision ResNet.conv1 under
 * models/tanet_models/tanet.py:125-150, as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).
 * d_x [N, 3, H, W] (W % 4 == 0); d_wp = the [64, 3, 7, 7] parameter packed [148][64]: row t = (c * 7 + kh) * 7 + kw holds
 * w[:, c, kh, kw], row 147 is zero; d_y [N, 64, OH, OW] raw convolution output, OH = (H - 1) / 2 + 1 (the NCHW tensor
 * vitta_stem_bn_relu_pool_* read).
 * -------------------------------------------------------------------------- */
int vitta_stem_conv7_f32(const float* d_x, const float* d_wp, int64_t N, int32_t H, int32_t W, float* d_y, void* stream);

/* --------------------------------------------------------------------------
 * A8 -- classification head: y[m][n] = b[n] + sum_k x[m][k] w[n][k]  (new_fc = nn.Linear(2048, num_class) on the pooled
 * per-frame features, models/tanet_models/tanet.py:105-123, 243-251).  M = frames (small), K % 4 == 0.
 * Backward: d_dx [M, K] (overwritten; NULL: skipped); d_dw [N, K] / d_db [N] are ACCUMULATED (may point into .grad /
 * the flat gradient arena; NULL: skipped, the frozen head of affine-only adaptation).
 * -------------------------------------------------------------------------- */
int vitta_linear_fwd_f32(const float* d_x, const float* d_w, const float* d_b, int64_t M, int32_t N, int32_t K, float* d_y,
                         void* stream);
int vitta_linear_bwd_f32(const float* d_dy, const float* d_x, const float* d_w, int64_t M, int32_t N, int32_t K, float* d_dx,
                         float* d_dw, float* d_db, void* stream);

/* The head of the ADAPTATION pass as two launches (round 5): segment consensus, view logits, prediction consistency and the
 * video logits in one forward launch; their whole backward down to the gradient of the pooled features in one more.
 * Replaces, at the reference's call sites, dropout's consumer chain `new_fc -> consensus -> compute_pred_consis -> mean(1)`
 * (models/tanet_models/tanet.py:243-251, corpus/basics.py:640-668, utils/pred_consistency_utils.py:15-31).
 *   d_y [B V T][D]: per-frame features AFTER dropout, row (b V + v) T + t; d_w [K][D], d_b [K] (or NULL): new_fc.
 *   forward : d_ybar [B V][D] (or NULL) = mean over the T segments; d_view_logits [B V][K] = ybar w^T + b (scratch the last
 *             workgroup re-reads: written through); d_out [B][K] = mean over the views; d_loss [1] = compute_pred_consis summed
 *             over the videos; d_gradc [B V][K] = d loss / d view logits.  d_ticket: 4 bytes, zero at rest (left zero).
 *   backward: dl = g_loss[0] * gradc + g_out[b] / V (d_g_loss, d_g_out device pointers or NULL = 0);
 *             d_dfeat [B V T][D] = (dl w)[b v] * scale, zeroed where d_mask (dropout's keep mask, one byte per element, or NULL)
 *             is 0 -- scale = 1 / (T (1 - p)); trainable head: d_dw [K][D] / d_db [K] ACCUMULATED (needs d_ybar and the
 *             scratch d_dl [B V][K]); NULL: skipped.
 * Supported: D % 4 == 0, B V <= 8, vitta_tanet_head_lds_bytes(...) <= 128 KB. */
size_t vitta_tanet_head_lds_bytes(int32_t B, int32_t V, int32_t K, int32_t D);
int vitta_tanet_head_fwd_f32(const float* d_y, const float* d_w, const float* d_b, int32_t B, int32_t V, int32_t T, int32_t K, int32_t D,
                             float* d_ybar, float* d_view_logits, void* d_ticket, float* d_out, float* d_loss, float* d_gradc,
                             void* stream);
int vitta_tanet_head_bwd_f32(const float* d_gradc, const float* d_g_loss, const float* d_g_out, const float* d_w, const void* d_mask,
                             float scale, int32_t B, int32_t V, int32_t T, int32_t K, int32_t D, float* d_dfeat, const float* d_ybar,
                             float* d_dl, float* d_dw, float* d_db, void* stream);
/* total loss of a step, corpus/basics.py:668: d_out[0] = la * d_a[0] + lb * d_b[0] (d_b NULL: 0); the same launch leaves the two
 * upstream gradients for d out = 1 in d_ga1[0] = la, d_gb1[0] = lb (NULL: skipped) -- `loss.backward()` then needs no launch here.
 * Its backward for any other d out in one launch: d_ga[0] = la * g, d_gb[0] = lb * g (d_g NULL: g = 1; d_gb NULL: skipped). */
int vitta_loss_axpby_f32(const float* d_a, const float* d_b, float la, float lb, float* d_out, float* d_ga1, float* d_gb1, void* stream);
int vitta_loss_axpby_bwd_f32(const float* d_g, float la, float lb, float* d_ga, float* d_gb, void* stream);

/* --------------------------------------------------------------------------
 * A10 -- dense layers of Video Swin-B: d_y[m][n] = epi(sum_k d_a[m][k] d_b[n][k]), all row-major fp32, K % 32 == 0
 * (the qkv / proj Linear of WindowAttention3D, models/videoswintransformer_models/swin_transformer.py:144, 165;
 * Mlp.fc1 + GELU + fc2, :30-35; PatchMerging.reduction, :304-311; replaces torch.nn.functional.linear / nn.GELU
 * there).  d_b is nn.Linear's own [out][in] weight; a data gradient is the same product with d_a = dy and d_b = the
 * transposed weight [in][out].  Exact fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32), fp32 accumulation.
 *   mode 0: y = acc (+ d_bias[n], NULL: none)
 *   mode 1: h = acc + d_bias[n]; d_pre[m][n] = h (NULL: not kept); y = 0.5 h (1 + erf(h / sqrt 2))      (nn.GELU)
 *   mode 2: y = acc * gelu'(d_aux[m][n])                 (fc2's data gradient arriving at fc1's output, d_aux = h)
 * tile: 0 = library's choice, 1 = 128 x 128 output tiles, 2 = 64 x 128, 3 = 64 x 64.  VITTA_ERR_UNSUPPORTED: K % 32 != 0 or an
 * operand of 2 GiB or more (vitta_gemm_nt_supported).
 * -------------------------------------------------------------------------- */
int vitta_gemm_nt_supported(int64_t M, int32_t N, int32_t K);
int vitta_gemm_nt_f32(const float* d_a, const float* d_b, const float* d_bias, const float* d_aux, float* d_y, float* d_pre,
                      int64_t M, int32_t N, int32_t K, int32_t mode, int32_t tile, void* stream);
/* The same product (64 x 64 tiles) as a STREAM-K launch of `grid` workgroups (a multiple of the CU count): the K slabs of all
 * tiles are cut into `grid` equal contiguous ranges, partial tiles meet in the workspace and the last arriver of a tile's ticket
 * adds them in range order (deterministic) and runs the epilogue.  For products whose tile count quantises badly on 256 CUs
 * (Swin-B stage 2, N = 512: 392 tiles).  d_workspace: vitta_gemm_nt_sk_workspace_bytes(grid) bytes, 16-byte aligned, its first
 * 256 KiB ZERO when first used (the kernel leaves them zero), one per stream. */
int64_t vitta_gemm_nt_sk_workspace_bytes(int32_t grid);
int vitta_gemm_nt_sk_f32(const float* d_a, const float* d_b, const float* d_bias, const float* d_aux, float* d_y, float* d_pre,
                         int64_t M, int32_t N, int32_t K, int32_t mode, int32_t grid, void* d_workspace, int64_t workspace_bytes,
                         void* stream);
/* The same product on bf16 MFMA operands (v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogues; an extension beside
 * BASELINE config 5's bf16 window attention, opt-in because the reference computes in fp32): d_a stays fp32 in memory and
 * is rounded to bf16 (nearest even) while it is staged, d_b_bf16 is the caller's bfloat16 copy of the weight [N][K].
 * K % 64 == 0. */
int vitta_gemm_nt_bf16w_f32(const float* d_a, const uint16_t* d_b_bf16, const float* d_bias, const float* d_aux, float* d_y,
                            float* d_pre, int64_t M, int32_t N, int32_t K, int32_t mode, int32_t tile, void* stream);

/* The dense product on bfloat16 operands IN MEMORY (gemm_bf16x.hip; the bf16 recipe of BASELINE config 5,
 * swin_transformer.py:30-35, 144, 165, 304-311 under recognizer3d.py:36-40's fp16 / bf16 wrapper):
 *   y[m][n] = sum_k a[m][k] b[n][k] (+ bias[n]),   d_a [M][K] bf16, d_b [N][K] bf16 (nn.Linear's own layout), d_y [M][N] fp32.
 * Both operands reach LDS by LDS-DMA, 128 x 128 tiles, 32-wide k-steps in a three-stage ring.  N % 128 == 0, K % 32 == 0 (vitta_gemm_bf16x_supported). */
int vitta_gemm_bf16x_supported(int64_t M, int64_t N, int64_t K);
int vitta_gemm_nt_bf16x_f32(const void* d_a, const void* d_b, const float* d_bias, float* d_y, int64_t M, int64_t N, int64_t K,
                            void* stream);
/* The same kernel with the epilogues of vitta_gemm_nt_f32 and 2-byte activations on either side (the bf16 recipe's data flow,
 * swin_transformer.py:30-35, 144, 165): mode 0 y = acc + bias; mode 1 h = acc + bias -> d_pre (bf16, optional), y = gelu(h);
 * mode 2 y = acc * gelu'(d_aux) with d_aux the bf16 pre-activation.  out_bf16: d_y is bfloat16 [M][N], else float32. */
int vitta_gemm_nt_bf16x(const void* d_a, const void* d_b, const float* d_bias, const void* d_aux, void* d_y, void* d_pre, int64_t M,
                        int64_t N, int64_t K, int32_t mode, int32_t out_bf16, void* stream);


/* --------------------------------------------------------------------------
 * N1 -- decoded RGB frames -> network input, bit-identical to the reference's PIL pipeline
 * (models/tanet_models/transforms.py:277-384 per-view multi-scale crop + Image.resize(BILINEAR); :46-54, :170-184
 * short-edge scale + centre crop; :637-678 stack + ToTorchFormatTensor(div 255); :140-152 GroupNormalize).
 * d_frames [n_frames][in_h][in_w][3] bytes (decoder layout); frame f belongs to view f / frames_per_view.  Per view:
 * d_origin (x0, y0) of its crop; per OUTPUT column x the taps of Pillow's 8-bit resampler: d_xbounds[v][x] = (first
 * input column relative to the crop, tap count), d_xcoef[v][x][0..kx) 22-bit fixed-point weights; same for rows.  The
 * horizontal pass is rounded to bytes before the vertical pass (as Pillow does).  d_lut [3][256] = the normalised fp32
 * value of each byte per channel.  d_out [n_frames * 3][out_h][out_w] (frames stacked on the channel axis).
 * kx / ky = row strides of the weight tables >= every window's tap count, zero padded (4 / 4 selects the unrolled path).
 * One workgroup = one frame x tile_rows output rows; lds_rows >= the input rows any tile's taps span (the caller built
 * the tables and knows; rows beyond it are dropped, never written out of bounds); lds_rows * 3 * out_w <= 64 KiB.
 * -------------------------------------------------------------------------- */
int vitta_frames_resample_norm_f32(const uint8_t* d_frames, int32_t n_frames, int32_t in_h, int32_t in_w,
                                   int32_t frames_per_view, const int32_t* d_origin, const int32_t* d_xbounds,
                                   const int32_t* d_xcoef, int32_t kx, const int32_t* d_ybounds, const int32_t* d_ycoef,
                                   int32_t ky, const float* d_lut, float* d_out, int32_t out_h, int32_t out_w,
                                   int32_t tile_rows, int32_t lds_rows, void* stream);

/* N1, Video Swin pipeline: cv2.resize(INTER_LINEAR) on uint8 frames as mmcv.imresize runs it
 * (models/videoswintransformer_models/transforms_backup.py:193-349 Resize; RandomResizedCrop / CenterCrop; Normalize;
 * FormatShape NCTHW), restated from OpenCV's resize.cpp -- cv2 is absent from this image, parity with cv2 itself is
 * UNPINNED.  d_frames [n_frames][in_h][in_w][3]; mode 0: copy of the out_h x out_w crop at (x0, y0); 1: the exact-2x area
 * shortcut on the 2 out_h x 2 out_w crop; 2: fixed-point bilinear, d_tables int32 [x0 | x1 | a0 | a1] (out_w each) then
 * [y0 | y1 | b0 | b1] (out_h each): absolute source indices and 11-bit weights (vitta_amd/frames.py::cv2_linear_axis).
 * Exactly one output: d_out_u8 [n_frames][out_h][out_w][3], or d_out_f32 [n_frames / clip_len][3][clip_len][out_h][out_w]
 * = (value - d_mean[c]) * d_stdinv[c]. */
int vitta_frames_cv2_resize(const uint8_t* d_frames, int32_t n_frames, int32_t in_h, int32_t in_w, int32_t x0, int32_t y0,
                            int32_t mode, const int32_t* d_tables, int32_t out_h, int32_t out_w, uint8_t* d_out_u8,
                            float* d_out_f32, int32_t clip_len, const float* d_mean, const float* d_stdinv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITTA_HIP_H_ */
