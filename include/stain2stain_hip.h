/* stain2stain_hip.h -- C ABI of libstain2stain_hip.so (gfx950 / MI355X only).
 *
 * The reference (nirschl-lab/stain2stain) has no FFI: its hot path is stock torch.nn layers selected
 * through Hydra `_target_` strings (configs/model/*.yaml).  These entry points are therefore
 * build-defined; each one names the reference computation it replaces.  The Python classes in
 * stain2stain_amd/ (same constructor signatures and state_dict keys as the reference's
 * SharedEncoder / FlowMatchingDecoder / TimeEmbedding) bind them through ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative S2S_ERR_* code; nothing throws across the ABI
 *   - all pointers are DEVICE pointers unless stated otherwise; the caller owns every buffer,
 *     kernels never allocate; `stream` is a hipStream_t passed as void*
 *   - dtype: S2S_BF16 (0) = bf16 tensors, bf16 MFMA, fp32 accumulate;
 *            S2S_F32 (1)  = fp32 tensors; MFMA products formed from a 3-way bf16 split (6 MFMAs, fp32-grade)
 *   - activation tensors are NHWC "views": element (n,h,w,c) lives at base[((n*H+h)*W+w)*ld + c];
 *     ld (pixel stride, elements) lets a channel slice of a wider buffer be used in place.
 *     Channel counts and ld must be multiples of 8, base pointers 16-byte aligned.
 *   - the image tensors at the network boundary are NCHW contiguous fp32 (the reference's layout).
 */
#ifndef STAIN2STAIN_HIP_H
#define STAIN2STAIN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define S2S_OK 0
#define S2S_ERR_SHAPE (-1)
#define S2S_ERR_ALIGN (-2)
#define S2S_ERR_DTYPE (-3)
#define S2S_ERR_LAUNCH (-4)
#define S2S_ERR_NULL (-5)

#define S2S_BF16 0
#define S2S_F32 1

/* ---- 3x3 convolution, MFMA implicit GEMM (conv3x3_mfma.hip) -------------------------------------
 * nn.Conv2d(k=3, padding=1) of DoubleConv: src/models/components/shared_encoder.py:15,18,
 * task_decoders.py:15,18.  Input = channels [0,c0) of x0 followed by [0,c1) of x1 (x1 may be NULL with
 * c1 = 0): the torch.cat([skip, up], dim=1) of task_decoders.py:49 without the copy.
 * w_packed: s2s_pack_conv3x3 output (forward layout for the forward pass, dgrad layout + swapped channel
 * roles for the data gradient).  stat_part (optional): float[2][Cout][s2s_conv3x3_stat_blocks()] partial
 * (sum, sum of squares) of the stored values per output channel and producing workgroup, for BatchNorm
 * (channel-major, the layout s2s_bn_finalize reads).  ep_scale/ep_shift
 * (optional, both or neither) and relu: y = relu?((acc + bias) * scale + shift) -- eval-mode BatchNorm
 * folded into the epilogue. */
int s2s_conv3x3_stat_blocks(int dtype, int B, int H, int W, int Cout);
int s2s_conv3x3_nhwc(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                     const void* w_packed, const float* bias, void* y, int ldy, float* stat_part,
                     const float* ep_scale, const float* ep_shift, int relu, int B, int H, int W, int Cout,
                     void* stream);

/* Diagnostic: with S2S_CONV_DBG=64 in the environment the bf16 kernel's workgroups record the shader-clock counter and
 * the 100 MHz wall clock at entry and exit; this copies {clk0, clk1, wall0, wall1} of the first n <= 8192 workgroups of
 * the last launch into a HOST buffer (scripts/clk_probe.py turns them into the MHz the kernel ran at). */
/* The same with a split-K workspace (optional): kwork = float[s2s_conv3x3_ksplit()][B*H*W][Cout]; when the launch
 * carries no statistics and has few output tiles (sampling one tile at a time) the 32-channel chunks are shared by
 * several workgroups and a second launch folds the fp32 partial tiles (+ bias, folded affine, ReLU). */
int s2s_conv3x3_ksplit(int dtype, int B, int H, int W, int Cout, int cin);
int s2s_conv3x3_nhwc_k(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1,
                       const void* w_packed, const float* bias, void* y, int ldy, float* stat_part,
                       const float* ep_scale, const float* ep_shift, int relu, float* kwork, int B, int H, int W,
                       int Cout, void* stream);
/* Rows of BatchNorm partial sums a training-step launch of these operands writes (>= 1): s2s_conv3x3_stat_blocks() of
 * them, or one per (workgroup, wave row) when the launch runs on a persistent kernel that carries the sums over a
 * workgroup's tiles in registers (the staged kernel; the per-tap persistent kernel with a single channel tile).  Size
 * stat_part as float[2][Cout][rows] and hand `rows` to s2s_conv3x3_nhwc_s (stat_rows = 0: the per-tile count). */
int s2s_conv3x3_stat_rows(int dtype, int B, int H, int W, int Cout, int c0, int c1, int ld0, int ld1, int has_bias);
/* Which kernel family a bias-free training-step launch of these operands runs on: 0 = the per-tap kernels (one barrier
 * per filter tap), 1 = conv3x3_stage_kernel streaming its weights per 32-channel chunk, 2 = the same with the filter
 * resident in LDS (<= 64 input channels).  Diagnostic: tests and scripts assert the path they mean to measure. */
int s2s_conv3x3_staged(int dtype, int B, int H, int W, int Cout, int c0, int c1, int ld0, int ld1, int ldy, int stats);
int s2s_conv3x3_nhwc_s(int dtype, const void* x0, int ld0, int c0, const void* x1, int ld1, int c1, const void* w_packed,
                       const float* bias, void* y, int ldy, float* stat_part, int stat_rows, const float* ep_scale,
                       const float* ep_shift, int relu, float* kwork, int B, int H, int W, int Cout, void* stream);
int s2s_debug_conv_clock(long* out_host, int n);

/* ---- 3x3 convolution weight gradient (conv3x3_wgrad_mfma.hip) -----------------------------------
 * autograd's conv2d weight gradient for the same layers.  part: float[splits][9][Cout][c0+c1] scratch;
 * grad_oihw: float[Cout][c0+c1][3][3], overwritten or accumulated. */
int s2s_conv3x3_wgrad_splits(int dtype, int B, int H, int W, int Cin, int Cout);
int s2s_conv3x3_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x0, int ld0, int c0,
                           const void* x1, int ld1, int c1, float* part, float* grad_oihw, int accumulate,
                           int B, int H, int W, void* stream);
/* the same in two launches: phases = 1 the split MFMA kernel (slabs into part), 2 the fold into grad_oihw, 3 both */
int s2s_conv3x3_wgrad_phase(int dtype, const void* dy, int lddy, int Cout, const void* x0, int ld0, int c0,
                            const void* x1, int ld1, int c1, float* part, float* grad_oihw, int accumulate,
                            int B, int H, int W, int phases, void* stream);

/* ---- stem and head (conv_edge.hip) ----------------------------------------------------------------
 * stem: first conv of SharedEncoder.inc (shared_encoder.py:15,67), NCHW fp32 image (Cin <= 6: RGB, or RGB + the
 *       mask channel of conditional_flow_matching_conditional_mask.py:62-64) -> NHWC.
 * head: FlowMatchingDecoder.outc / SegmentationDecoder.outc, Conv2d(k=1) to Cout <= 8 (task_decoders.py:100,132,169),
 *       NHWC -> NCHW fp32; the fused head+loss entry point takes Cout <= 4. */
int s2s_stem_stat_blocks(int B, int H, int W);
int s2s_stem_conv3x3_fwd(int dtype, const float* x_nchw, const float* w_oihw, const float* bias, void* y, int ldy,
                         float* stat_part, int B, int H, int W, int Cin, int Cout, void* stream);
int s2s_stem_wgrad_blocks(int B, int H, int W);
/* part: float[ceil(Cin/3)][2*blocks][Cout][32]; dbias may be NULL */
int s2s_stem_conv3x3_wgrad(int dtype, const void* dy, int lddy, const float* x_nchw, float* part, float* dw_oihw,
                           float* dbias, int accumulate, int B, int H, int W, int Cin, int Cout, void* stream);
int s2s_head_conv1x1_fwd(int dtype, const void* x, int ldx, const float* w, const float* bias, float* y_nchw, int B,
                         int H, int W, int C, int Cout, void* stream);
int s2s_head_wgrad_blocks(int B, int H, int W);
/* part: float[blocks][Cout][C+1] */
int s2s_head_conv1x1_bwd(int dtype, const float* dy_nchw, const void* x, int ldx, const float* w, void* dx, int lddx,
                         float* part, float* dw, float* dbias, int accumulate, int B, int H, int W, int C, int Cout,
                         void* stream);

/* head + loss + their backward in one pass over the last activation (training step only):
 * v = W a + b (optional output), loss = mean((v-u)^2), dx = W^T dv, dW/db with dv = 2 (v-u) grad_scale / count
 * (task_decoders.py:132 + conditional_flow_matching.py:72).  part: float[blocks][Cout][C+1]; lpart: double[blocks] */
int s2s_head_loss_blocks(int B, int H, int W);
int s2s_head_loss_fused(int dtype, const void* x, int ldx, const float* w, const float* bias, const float* u_nchw,
                        float* v_nchw, void* dx, int lddx, float grad_scale, float* part, double* lpart, float* dw,
                        float* dbias, float* loss, int accumulate, int B, int H, int W, int C, int Cout, void* stream);

/* ---- BatchNorm2d + ReLU (+ MaxPool2d(2)) (norm_act.hip) -------------------------------------------
 * nn.BatchNorm2d(eps 1e-5, momentum 0.1) -> nn.ReLU of DoubleConv (shared_encoder.py:16-20) and the
 * nn.MaxPool2d(2) of the following Down block (shared_encoder.py:33). */
int s2s_bn_finalize(const float* part, int nblk, int C, long count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long* num_batches_tracked, float momentum, float eps,
                    float* mean, float* invstd, float* scale, float* shift, void* stream);
/* the same when the statistics were taken of the convolution WITHOUT its bias (conv_bias: [C] or NULL): mean / invstd /
 * scale / shift refer to the unbiased conv output the caller stores, the running mean is that of conv + bias, as
 * nn.BatchNorm2d behind nn.Conv2d(bias=True) keeps it (shared_encoder.py:15-16) */
int s2s_bn_finalize_b(const float* part, int nblk, int C, long count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, long* num_batches_tracked, float momentum, float eps,
                      float* mean, float* invstd, float* scale, float* shift, const float* conv_bias, void* stream);
int s2s_bn_eval_prepare(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift, void* stream);
/* y = relu(x*scale+shift); pool (optional) = 2x2/stride-2 max of y, floor mode */
int s2s_bn_relu_apply(int dtype, const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy,
                      void* pool, int ldp, int B, int H, int W, int C, void* stream);
int s2s_maxpool2(int dtype, const void* x, int ldx, void* pool, int ldp, int B, int H, int W, int C, void* stream);
int s2s_bn_bwd_blocks(int B, int H, int W, int C);
/* g1: gradient wrt the ReLU output (may be NULL if gp given); gp: gradient wrt the pooled output (may be
 * NULL); scale/shift: the forward's folded affine (the ReLU output is recomputed from x, not read);
 * x: saved conv output; work: float[4*blocks*C + 2*C];
 * dbias_conv (optional): gradient of the preceding conv's bias = per-channel sum of dx, which the BatchNorm backward
 * makes identically zero; written as exact 0 (the reference's autograd sums rounding noise there).  With
 * S2S_BN_DBIAS_SUM=1 in the environment the sum is formed instead. */
int s2s_bn_relu_bwd(int dtype, const void* g1, int ldg1, const void* gp, int ldgp, const float* scale,
                    const float* shift, const void* x, int ldx, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                    float* dbeta, float* dbias_conv, int accumulate, void* dx, int lddx, float* work, int B, int H,
                    int W, int C, void* stream);
/* SyncBatchNorm (Lightning's sync_batchnorm: True, configs/trainer/ddp.yaml:9 = torch.nn.SyncBatchNorm): statistics
 * over the GLOBAL batch.  Forward: s2s_bn_partial_sums folds the conv epilogue's partials into sums[2][C] (sum, sum of
 * squares), the host all-reduces them and calls s2s_bn_finalize on them (nblk = 1, count = global element count).
 * Backward: s2s_bn_relu_bwd_phase(phases = 1, count_total = global count) leaves (sum_dy, sum_dy_xhat) / count_total
 * in work[4 nb C .. + 2 C), the host all-reduces those 2 C floats, phases = 2 applies; phases = 3 with the local
 * count is s2s_bn_relu_bwd.  dgamma / dbeta stay rank-local sums (the gradient all-reduce averages them), as in torch. */
int s2s_bn_partial_sums(const float* part, int nblk, int C, float* sums, void* stream);
int s2s_bn_relu_bwd_phase(int dtype, const void* g1, int ldg1, const void* gp, int ldgp, const float* scale,
                          const float* shift, const void* x, int ldx, const float* mean, const float* invstd,
                          const float* gamma, float* dgamma, float* dbeta, float* dbias_conv, int accumulate, void* dx,
                          int lddx, float* work, int B, int H, int W, int C, long count_total, int phases, void* stream);

/* ---- bilinear x2, align_corners=True (+ F.pad to the skip size) (resample.hip) --------------------
 * nn.Upsample + F.pad of Up.forward (task_decoders.py:34,42-47); bias_nc (optional, float[B][C]) is added
 * to the input first: x = bottleneck + t[:, :, None, None] (task_decoders.py:119-125). */
int s2s_upsample2x_bilinear_ac_fwd(int dtype, const void* x, int ldx, const float* bias_nc, void* y, int ldy, int B,
                                   int Hin, int Win, int Hout, int Wout, int C, void* stream);
int s2s_upsample2x_bilinear_ac_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx, int B, int Hin, int Win,
                                   int Hout, int Wout, int C, void* stream);
int s2s_pixel_sum(int dtype, const void* x, int ldx, float* out_nc, int B, int HW, int C, int accumulate,
                  void* stream);

/* ---- flow matching glue (flow.hip) --------------------------------------------------------------- */
/* TimeEmbedding.forward, shared_encoder.py:114-135 */
int s2s_time_embedding(const float* t, float* out, int B, int dim, void* stream);
/* nn.Linear / nn.SiLU of FlowMatchingDecoder.time_mlp, time_proj (task_decoders.py:82-89) */
int s2s_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, void* stream);
int s2s_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int accumulate,
                   int B, int K, int N, void* stream);
int s2s_silu_fwd(const float* h, float* a, int n, void* stream);
int s2s_silu_bwd(const float* h, const float* da, float* dh, int n, void* stream);
/* xt = t*x1 + (1-t)*x0 + sigma*eps, ut = x1 - x0 (conditional_flow_matching.py:66; torchcfm 1.0.7) */
int s2s_cfm_sample(const float* x0, const float* x1, const float* t, const float* eps, float sigma, float* xt,
                   float* ut, int B, long per_sample, void* stream);
/* loss = mean((v-u)^2) (conditional_flow_matching.py:72); dv optional; work: double[1024] */
int s2s_mse_loss(const float* v, const float* u, float* dv, float grad_scale, float* loss, double* work, long count,
                 void* stream);
/* Scaled RMS error of an embedded Runge-Kutta step, the step-size control of the adaptive sampler that stands in for
 * torchdyn's dopri5 (conditional_flow_matching.py:157-170): out[0] = sqrt(mean((e / (atol + rtol max(|y0|,|y1|)))^2));
 * work: double[1024]. */
int s2s_ode_error_norm(const float* e, const float* y0, const float* y1, float atol, float rtol, float* out,
                       double* work, long n, void* stream);
int s2s_axpy(float* x, const float* y, float a, long n, void* stream);
int s2s_fill_f32(float* x, float v, long n, void* stream);
/* t[0..n) <- table[*counter], then *counter += 1: the time input of a hipGraph-captured Euler step (the captured
 * kernel walks a table of the node times, so a replay needs no host-side argument). */
int s2s_euler_tick(float* t, int n, const float* table, int* counter, void* stream);

/* ---- optimiser, packing, layout (optim.hip) ------------------------------------------------------ */
/* torch.optim.Adam step over a flat fp32 buffer (configs/model/*.yaml:3-7) */
int s2s_adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, void* stream);
/* the same step with its scalars in DEVICE memory, hyper[8] = {lr, beta1, beta2, eps, weight_decay, 1 - beta1^step,
 * sqrt(1 - beta2^step), grad_scale}: the launch a hipGraph-captured training step replays while the host refreshes the
 * eight floats (Lightning's optimizer.step() / scheduler surface, conditional_flow_matching.py:112-131) */
int s2s_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, void* stream);
/* The same step over a list of tensors in one launch (ordinary module parameters + the gradients autograd left in .grad:
 * stain2stain_amd.FusedAdam, the optimiser of the drop-in path): desc = device long[ntensors][6] {p, g, m, v, n, first
 * block}; a tensor occupies s2s_adam_multi_blocks(n) blocks, total = their sum. */
long s2s_adam_multi_blocks(long n);
int s2s_adam_multi(const void* desc, int ntensors, long total, int step, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, void* stream);
long s2s_pack_conv3x3_fwd_elems(int Cout, int Cin);
long s2s_pack_conv3x3_dgrad_elems(int Cout, int Cin);
int s2s_pack_conv3x3(int dtype, const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, void* stream);
/* every conv layer in one launch; desc = device int64[nlayers][6] {w_oihw, w_fwd, w_dgrad, Cout, Cin, first 32x32 tile};
 * total = number of 32x32 (co,ci) tiles over all layers */
int s2s_pack_conv3x3_batched(int dtype, const void* desc, int nlayers, long total, void* stream);
int s2s_nchw_to_nhwc(int dtype, const float* x_nchw, void* y, int ldy, int B, int C, int H, int W, void* stream);
int s2s_nhwc_to_nchw(int dtype, const void* x, int ldx, float* y_nchw, int accumulate, int B, int C, int H, int W,
                     void* stream);

/* ---- paired input pipeline (input_pipeline.hip) -- SURVEY section 8 row f1 --------------------------
 * PairedDataset.__getitem__ transform chain (src/data/paired_data_module.py:170-199): shared crop + h/v flips,
 * to_tensor, Normalize(0.5, 0.5).  src/tgt: uint8 [B][Hs][Ws][3]; params: int32 [B][4] {top, left, hflip, vflip};
 * out: float [B][3][S][S]. */
int s2s_paired_crop_flip_normalize(const void* src_u8, const void* tgt_u8, const int* params, float* out_src,
                                   float* out_tgt, int B, int Hs, int Ws, int S, void* stream);
/* The use_augmentation=False branch (paired_data_module.py:200-211): TF.resize on a PIL image =
 * PIL.Image.resize((Wo, Ho), BILINEAR) -- Pillow's two-pass 22-bit fixed-point resampler with a uint8 intermediate
 * (Pillow src/libImaging/Resample.c) -- then to_tensor and Normalize(0.5, 0.5).  The coefficient tables (bounds:
 * int32 [out][2] = first input index, tap count; kk: int32 [out][ksize]) are Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc, built on the host (stain2stain_amd/data.py).  src: uint8 [B][Hs][Ws][3]; tmp: uint8
 * [B][Hs][Wo][3] scratch; out_u8 (optional): uint8 [B][Ho][Wo][3]; out_f (optional): float [B][3][Ho][Wo]. */
int s2s_pil_resize_bilinear_normalize(const void* src_u8, void* tmp_u8, const int* bounds_h, const int* kk_h,
                                      int ksize_h, const int* bounds_v, const int* kk_v, int ksize_v, void* out_u8,
                                      float* out_f, int B, int Hs, int Ws, int Ho, int Wo, void* stream);

/* ---- segmentation loss (seg_loss.hip) -- SURVEY section 8 row f2 --------------------------------------
 * seg = dw * DiceLoss(sigmoid(z), g) + (1-dw) * BCEWithLogits(z, g)  (conditional_flow_matching_multitask.py:36-53,
 * 174-202).  z, g: float[n]; out: float[3] = {seg, dice, bce}; dz (optional) = grad_scale * d seg / dz;
 * work: double[512*4 + 4]. */
int s2s_seg_loss(const float* z, const float* g, float* dz, float* out, double* work, long n, float smooth,
                 float dice_weight, float grad_scale, void* stream);
/* Multiclass form: dw * MulticlassDiceLoss(softmax(z), t) + (1-dw) * CrossEntropy(z, t, ignore_index)
 * (conditional_flow_matching_multitask_multiclassloss.py:41-83, 159, 214-245).  z: float[B][C][HW] logits,
 * target: int64[B][HW]; out: float[3] = {seg, dice, ce}; dz (optional) float[B][C][HW];
 * work: double[(512 + 1) * 26]; 2 <= C <= 8. */
int s2s_seg_loss_multiclass(const float* z, const long* target, float* dz, float* out, double* work, long B, long HW,
                            int C, int ignore_index, float smooth, float dice_weight, float grad_scale, void* stream);

/* ---- loss variants and class conditioning (loss_variants.hip) -- SURVEY section 8 row f4 -----------------
 * ROI-weighted MSE (conditional_flow_matching_masked.py:76-90): w = 1 + roi_lambda * mask,
 *   out[0] = sum(w (v-u)^2) / (sum(w) + 1e-8), w broadcast over C.  v, u: float[B][C][HW]; mask: float[B][HW];
 *   dv optional; work: double[512*2 + 2].
 * ROI Charbonnier (conditional_flow_matching_ROI_loss.py:78-95): out[0] = sum(sqrt((p-t)^2 + eps_charb^2) * mask)
 *   / (sum(mask) * C + eps_area); value only (its inputs are data).
 * Class conditioning (class_conditional_flow_matching.py:39-71, build-defined network side): out = temb + table[y]. */
int s2s_weighted_mse(const float* v, const float* u, const float* mask, float* dv, float* out, double* work, long B,
                     int C, long HW, float roi_lambda, float grad_scale, void* stream);
int s2s_charbonnier_roi(const float* pred, const float* truth, const float* mask, float* out, double* work, long B,
                        int C, long HW, float eps_charb, float eps_area, void* stream);
int s2s_class_embed_add(const float* temb, const float* table, const long* y, float* out, int B, int dim,
                        void* stream);
int s2s_class_embed_bwd(const float* dout, const long* y, float* dtable, int accumulate, int B, int dim,
                        int num_classes, void* stream);

/* ---- InstanceNorm2d + LeakyReLU, fused (instnorm.hip) ----------------------------------------------
 * Row a13 of SURVEY.md section 8 (pix2pix generator / PatchGAN discriminator named by BASELINE.json's north_star).
 * The reference repository has no such model, so these replace no reference line: the semantics are
 * torch's nn.InstanceNorm2d(C, eps, affine = gamma/beta given or both NULL, no running statistics) followed by
 * nn.LeakyReLU(slope) (slope 0 = ReLU), NHWC with a pixel stride, C % 8 == 0.
 * work: float[2*B*C*s2s_instnorm_blocks()] for the forward, that + 2*B*C for the backward.
 * stats: float[4][B][C] = mean, 1/std, scale, shift; written by the forward, read by the backward.
 * backward: dx = d/dx of sum(g * y); dgamma/dbeta (both or neither) receive or accumulate the affine gradients. */
int s2s_instnorm_blocks(int B, int H, int W, int C);
int s2s_instnorm_lrelu_fwd(int dtype, const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                           float* work, float* stats, int B, int H, int W, int C, float eps, float slope,
                           void* stream);
int s2s_instnorm_lrelu_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const float* stats, void* dx,
                           int lddx, float* dgamma, float* dbeta, int accumulate, float* work, int B, int H, int W,
                           int C, float slope, void* stream);

/* ---- 2x2-tap convolution on the MFMA loop (conv3x3_mfma.hip) -----------------------------------------
 * Row a13 again (no reference line to replace): nn.Conv2d(k=4, stride=2, padding=1) of a pix2pix
 * encoder / PatchGAN layer is a 2x2 "valid" convolution over the space-to-depth image of the zero-padded input
 * (4*Cin channels, no wasted MACs) - pad = 0, input (H+1) x (W+1) - and its data gradient, which is also
 * nn.ConvTranspose2d(k=4, stride=2, padding=1), is the same loop with flipped taps - pad = 1, input (H-1) x (W-1).
 * H, W: OUTPUT size.  w_packed: bf16 [ceil(cin/32)][tap a*2+b][Cout][32].  bf16 only so far.
 * stat_part (optional): float[2][Cout][s2s_conv2x2_stat_blocks()] as for the 3x3 kernel. */
int s2s_conv2x2_stat_blocks(int B, int H, int W, int Cout);
int s2s_conv2x2_nhwc(int dtype, const void* x, int ldx, int cin, const void* w_packed, const float* bias, void* y,
                     int ldy, float* stat_part, int B, int H, int W, int Cout, int pad, void* stream);
/* Layout change ahead of s2s_conv2x2_nhwc: xs[n][p][q][(r*2+s)*C + c] = xpad[n][2p+r][2q+s][c] with xpad = x plus a
 * one-pixel zero border; x [B][H][W][C] (H, W even), xs [B][H/2+1][W/2+1][4C].  inverse != 0 writes x from xs (the data
 * gradient's way back; the border cells are dropped).  bf16. */
int s2s_space_to_depth_pad1(int dtype, const void* x, int ldx, void* xs, int ldxs, int inverse, int B, int H, int W,
                            int C, void* stream);
/* out[c] (+)= sum over the npix pixels of an NHWC tensor (bias gradient of the 4x4 layers); work:
 * float[2*C*s2s_channel_sum_blocks()]. */
int s2s_channel_sum_blocks(long npix, int C);
int s2s_channel_sum(int dtype, const void* x, int ldx, float* work, float* out, long npix, int C, int accumulate,
                    void* stream);
/* fp32 master weight [Cout][Cin][4][4] of a 4x4 layer -> its two bf16 MFMA operands: wf (forward,
 * [ceil(K/32)][taps][Cout][32]) and wd (data gradient / transposed form, [ceil(Cout/32)][taps][K][32], taps flipped);
 * stride 2: taps = 4, K = 4*Cin (space-to-depth channel order); stride 1: taps = 16, K = Cin. */
int s2s_pack_conv4x4(const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int stride, void* stream);
/* nn.Conv2d(k=4, stride=1, padding=1) of the PatchGAN's last two layers (pad = 1, input (H+1) x (W+1)) and its data
 * gradient (pad = 2, input (H-1) x (W-1), taps flipped by the packing) on the same loop with 16 taps.
 * w_packed: bf16 [ceil(cin/32)][tap kh*4+kw][Cout][32]. */
int s2s_conv4x4s1_nhwc(int dtype, const void* x, int ldx, int cin, const void* w_packed, const float* bias, void* y,
                       int ldy, float* stat_part, int B, int H, int W, int Cout, int pad, void* stream);
/* weight gradient of the pad = 1 form: grad16[tap kh*4+kw][Cout][cin] (+)= sum over the batch, dY [B][H][W][Cout],
 * X [B][H+1][W+1][cin]; part: float[s2s_conv4x4s1_wgrad_splits()][16][Cout][cin] scratch. */
int s2s_conv4x4s1_wgrad_splits(int B, int H, int W, int Cin, int Cout);
int s2s_conv4x4s1_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x, int ldx, int cin,
                             float* part, float* grad16, int accumulate, int B, int H, int W, void* stream);
/* weight gradient of the pad = 0 form: grad2[tap a*2+b][Cout][cin] (+)= sum_{n,i,j} dY[n][i][j][:] x X[n][i+a][j+b][:],
 * dY [B][H][W][Cout], X [B][H+1][W+1][cin]; part: float[s2s_conv2x2_wgrad_splits()][4][Cout][cin] scratch. */
int s2s_conv2x2_wgrad_splits(int B, int H, int W, int Cin, int Cout);
int s2s_conv2x2_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x, int ldx, int cin, float* part,
                           float* grad2, int accumulate, int B, int H, int W, void* stream);

/* ---- the pix2pix G + D step's remaining kernels (row a13; no reference line to replace) -------------
 * General form of the two convolutions above, both dtypes (fp32 = three-way bf16 split, the parity mode):
 *   ks = 2, pad 0 | 1: 4x4 stride-2 convolution on the space-to-depth image | its data gradient = transposed convolution
 *   ks = 4, pad 1 | 2: 4x4 stride-1 convolution | its data gradient.          H, W: OUTPUT size.
 * w_packed: [ceil(cin/32)][ks*ks][Cout][32] in the activation dtype (s2s_pack_conv4x4_t).
 * act != 0: y = t > 0 ? t : act_slope * t on the bias-added output (the layers WITHOUT a norm: LeakyReLU(0.2) / ReLU).
 * y2 (optional, pixel stride ldy2): a second copy max(y, 0), the ReLU'd skip tensor written into the decoder's
 * concatenation buffer.  bias_mod (0 = Cout): the bias of output channel n is bias[n % bias_mod] -- Cout / 4 for the
 * transposed layers, whose four sub-pixel channel groups share the layer's bias vector.  stat_part (optional): float[2][Cout][s2s_convkxk_stat_blocks()]. */
int s2s_convkxk_stat_blocks(int dtype, int B, int H, int W, int Cout, int ks);
int s2s_convkxk_nhwc(int dtype, const void* x, int ldx, int cin, const void* w_packed, const float* bias, int bias_mod,
                     void* y, int ldy, void* y2, int ldy2, int act, float act_slope, float* stat_part, float* kwork, int B,
                     int H, int W, int Cout, int ks, int pad, void* stream);
/* Split-K for the small maps of the inner U-Net levels: with kwork = float[s2s_convkxk_ksplit()][B*H*W][Cout] (optional,
 * bf16 only) that many workgroups share an output tile, each reducing a range of the input-channel chunks, and a
 * second launch folds the fp32 partial tiles (+ bias, activation).  Returns 1 when the launch is not split. */
int s2s_convkxk_ksplit(int dtype, int B, int H, int W, int Cout, int cin, int ks);
/* Weight gradient of the pad-0 (ks = 2) / pad-1 (ks = 4) forms: dY [B][H][W][Cout], X [B][H+1][W+1][cin].
 * layout 0: grad[ks*ks][Cout][cin]; layout 1: nn.Conv2d's [Cout][C][4][4] (ks = 2: cin = 4 C over the space-to-depth
 * channels; with the roles of a transposed layer's input and output gradient exchanged the same call yields
 * nn.ConvTranspose2d's [Cin][Cout][4][4]).  part: float[s2s_convkxk_wgrad_splits()][ks*ks][Cout][cin] scratch. */
int s2s_convkxk_wgrad_splits(int dtype, int B, int H, int W, int Cin, int Cout, int ks);
int s2s_convkxk_wgrad_nhwc(int dtype, const void* dy, int lddy, int Cout, const void* x, int ldx, int cin, float* part,
                           float* grad, int layout, int accumulate, int B, int H, int W, int ks, int x_plain, void* stream);
/* The 4x4 stride-2 layers without a layout pass (bf16): s2s_conv4x4s2_nhwc reads the PLAIN input x [B][2H][2W][Cin]
 * (Cin a power of two; the loader does the space-to-depth in its addresses) -> y [B][H][W][Cout];
 * s2s_convt4x4s2_nhwc is the transposed layer / the convolution's data gradient by sub-pixel phase, x [B][h][w][Cin] ->
 * PLAIN y [B][2h][2w][C] (C % 64 == 0), wd = the data-gradient operand of s2s_pack_conv4x4.  kwork (optional):
 * float[..._ksplit()][output pixels][channels] for the split-K form of the small maps.  The weight gradient takes the
 * plain tensor with x_plain = 1 in s2s_convkxk_wgrad_nhwc (cin = 4 C stays the virtual channel count). */
int s2s_conv4x4s2_ksplit(int B, int H, int W, int Cout, int Cin);
int s2s_conv4x4s2_nhwc(int dtype, const void* x, int ldx, int Cin, const void* wf, const float* bias, void* y, int ldy,
                       void* y2, int ldy2, int act, float act_slope, float* kwork, int B, int H, int W, int Cout,
                       void* stream);
int s2s_convt4x4s2_ksplit(int B, int h, int w, int C, int Cin);
int s2s_convt4x4s2_nhwc(int dtype, const void* x, int ldx, int Cin, const void* wd, const float* bias, void* y, int ldy,
                        float* kwork, int B, int h, int w, int C, void* stream);
/* The same two layers on the generator's INNER levels as ONE launch with what follows them (conv_small.hip; bf16):
 * a workgroup owns whole samples x 16 output channels and all of K, so there is no split-K, no partial slab and no
 * reduce, and InstanceNorm + activation (forward) or their backward (data gradient) are the epilogue.
 *   mode 1: nn.Conv2d(4, 2, 1) from the plain x [B][2h][2w][Cin] with wf -> [B][h][w][Cout], h * w <= 64;
 *   mode 2: nn.ConvTranspose2d(4, 2, 1) / the convolution's data gradient from x [B][h][w][Cin] with wd ->
 *           [B][2h][2w][Cout], h * w <= 16.   h, w powers of two, Cin % 64 == 0, Cout % 16 == 0; s2s_convsm_ok() != 0 says a shape is taken.
 *   epi 0: y = act ? lrelu(conv + bias, slope) : conv + bias; y2 (optional) = relu(y).
 *   epi 1: raw (optional) = bf16(conv + bias); stats[4][B][Cout] = mean, invstd, invstd, -mean * invstd of raw per
 *          (sample, channel) [InstanceNorm2d(affine=False, eps)]; y = lrelu(norm(raw), slope); y2 (optional) = relu(norm).
 *   epi 2: g = conv (no bias).  Output channels [0, bwd_c0): y2 = g (pixel stride ldy2, channel index unchanged).
 *          Output channels [bwd_c0, Cout), as channel c - bwd_c0 of z / stats_in / g2 / y: the backward of epi 1 for
 *          that tensor, dzn = norm(z) > 0 ? g + g2 : slope * g, y = invstd * (dzn - mean(dzn) - xhat * mean(dzn xhat)). */
int s2s_convsm_ok(int dtype, int mode, int B, int h, int w, int Cin, int Cout);
int s2s_convsm_nhwc(int dtype, int mode, const void* x, int ldx, int Cin, const void* w_packed, const float* bias, int epi,
                    int act, float slope, float eps, void* raw, int ldraw, void* y, int ldy, void* y2, int ldy2,
                    float* stats, const float* stats_in, const void* z, int ldz, const void* g2, int ldg2, int bwd_c0,
                    int B, int h, int w, int Cout, void* stream);
/* InstanceNorm + LeakyReLU with a second output y2 = relu(z) (pixel stride ldy2, optional) and, backward, a second
 * incoming gradient g2 (wrt y2, optional): dz = z > 0 ? g + g2 : slope * g.  Otherwise as s2s_instnorm_lrelu_fwd/bwd. */
int s2s_instnorm_lrelu_fwd2(int dtype, const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                            void* y2, int ldy2, float* work, float* stats, int B, int H, int W, int C, float eps,
                            float slope, void* stream);
int s2s_instnorm_lrelu_bwd2(int dtype, const void* g, int ldg, const void* g2, int ldg2, const void* x, int ldx,
                            const float* stats, void* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                            float* work, int B, int H, int W, int C, float slope, void* stream);
/* The single-launch forms (bf16, no affine, maps of at most 32 x 32: s2s_instnorm_split_ok() != 0) with the input folded
 * from a split-K convolution's partial slabs, float[S][B*H*W][C] as s2s_conv4x4s2_nhwc / s2s_convt4x4s2_nhwc leave them in
 * kwork when called with y == NULL -- one launch instead of reduce + norm, bit-identical to the two-launch form.
 * fwd: raw (bf16, what the backward reads) <- fold + bias, then s2s_instnorm_lrelu_fwd2 on it; bwd: g <- fold. */
int s2s_instnorm_split_ok(int dtype, int H, int W, int C);
int s2s_instnorm_lrelu_fwd_split(int dtype, const float* slabs, int S, const float* bias, void* raw, int ldraw, void* y,
                                 int ldy, void* y2, int ldy2, float* stats, int B, int H, int W, int C, float eps,
                                 float slope, void* stream);
int s2s_instnorm_lrelu_bwd_split(int dtype, const float* gslabs, int S, const void* g2, int ldg2, const void* x, int ldx,
                                 const float* stats, void* dx, int lddx, int B, int H, int W, int C, float slope,
                                 void* stream);
/* s2s_pack_conv4x4 with the operands in the activation dtype; the batched form re-packs every 4x4 layer of a network in
 * one launch after the fused Adam step: desc = device long[nlayers][7] {w master, wf, wd, Cout, Cin, stride == 2, first
 * block}, a layer occupies s2s_pack_conv4x4_blocks() blocks, total = their sum. */
int s2s_pack_conv4x4_t(int dtype, const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int stride, void* stream);
long s2s_pack_conv4x4_blocks(int Cout, int Cin, int stride);
int s2s_pack_conv4x4_batched(int dtype, const void* desc, int nlayers, long total, void* stream);
/* out[B][H][W][8] (pixel stride ldo) <- [a (ca channels) | b (cb channels, optional) | zeros] from NCHW fp32 images: the
 * generator's input (source tile) and the discriminator's (source | target). */
int s2s_p2p_pack_input(int dtype, const float* a_nchw, int ca, const float* b_nchw, int cb, void* out, int ldo, int B,
                       int H, int W, void* stream);
/* The inverse for gradients: out_nchw[n][c][p] (fp32) = in[n][p][c0 + c], in = an 8-channel NHWC image (pixel stride ldi). */
int s2s_p2p_unpack(int dtype, const void* in, int ldi, int c0, int C, float* out_nchw, int B, int H, int W, void* stream);
/* Generator head: fake = tanh(h[..., :C]); d_in <- [src | fake | zeros] (the discriminator's input, 8 channels);
 * fake_nchw (optional) <- fake as NCHW fp32; l1_out[0] = mean |fake - tgt|.  work: double[s2s_p2p_tanh_l1_blocks()]. */
int s2s_p2p_tanh_l1_blocks(int B, int H, int W);
int s2s_p2p_tanh_l1_fwd(int dtype, const void* h, int ldh, const float* src_nchw, const float* tgt_nchw, void* d_in,
                        int ldd, float* fake_nchw, float* l1_out, void* work, int B, int H, int W, int C, void* stream);
/* dh[..., c] = (l1_scale * sign(fake - tgt) + gd[..., C + c]) * (1 - fake^2) for c < C, 0 on the padding channels;
 * gd (optional) = gradient wrt the discriminator's input. */
int s2s_p2p_tanh_l1_bwd(int dtype, const void* h, int ldh, const float* tgt_nchw, const void* gd, int ldg, float l1_scale,
                        void* dh, int lddh, int B, int H, int W, int C, void* stream);
/* BCEWithLogits of a PatchGAN logit map z[N][HW][ldz] (logit = channel 0): samples n < n_real against ones, the others
 * against zeros.  out2[0] / out2[1] = the two means; dz (optional, 8 channels, pixel stride lddz)[..., 0] =
 * w_real * (sigmoid(z) - 1) | w_fake * sigmoid(z), 0 on the padding channels. */
int s2s_p2p_bce_logits(int dtype, const void* z, int ldz, int n_real, float w_real, float w_fake, void* dz, int lddz,
                       float* out2, int N, int HW, void* stream);
/* The same spread over s2s_p2p_bce_blocks(N, HW) workgroups (one workgroup's softplus / sigmoid arithmetic for a
 * batch-64 map is 20-40 us); work: double[2 * s2s_p2p_bce_blocks(N, HW)], summed in workgroup order. */
int s2s_p2p_bce_blocks(int N, int HW);
int s2s_p2p_bce_logits_w(int dtype, const void* z, int ldz, int n_real, float w_real, float w_fake, void* dz, int lddz,
                         float* out2, double* work, int N, int HW, void* stream);
/* Backward of a LeakyReLU / ReLU without a norm in front, from its stored output a: dz = a > 0 ? g + g2 : slope * g
 * (g2 optional: the gradient wrt the ReLU'd copy); dbias (optional) (+)= per-channel sum of dz.
 * work: float[C * s2s_p2p_act_bwd_blocks()]. */
int s2s_p2p_act_bwd_blocks(long npix, int C);
int s2s_p2p_act_bwd(int dtype, const void* g, int ldg, const void* g2, int ldg2, const void* a, int lda, float slope,
                    void* dz, int lddz, float* work, float* dbias, int accumulate, long npix, int C, void* stream);

/* ---- streams (runtime.hip) -------------------------------------------------------------------------
 * A HIP stream confined to the compute units whose bits are set in mask_words (bit i of word i/32 = CU i, as
 * hipExtStreamCreateWithCUMask numbers them); *out_stream receives the hipStream_t.  Used for the weight-gradient side
 * stream of the fused trainers; any hipStream_t may be passed as `stream` to every entry point above. */
int s2s_stream_create_cu_mask(const unsigned* mask_words, int n_words, long* out_stream);
int s2s_stream_destroy(void* stream);
/* Events for timing brackets: timing_only != 0 creates the event with hipEventDisableSystemFence (a timestamp marker
 * without the system-scope fence of a default event); elapsed_ms wants both events complete. */
int s2s_event_create(int timing_only, long* out_event);
int s2s_event_record(void* event, void* stream);
int s2s_event_elapsed_ms(void* start, void* stop, float* out_ms);
int s2s_event_synchronize(void* event);
int s2s_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* STAIN2STAIN_HIP_H */
