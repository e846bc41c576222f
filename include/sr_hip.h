/*
 * sr_hip.h -- C ABI of libsr_hip.so, the MI355X (gfx950) implementation of the
 * FRVSR / TecoGAN generator's per-frame recurrent step.
 *
 * The reference (javiergonzalezya13/Super-Resolution) has no native boundary:
 * the hot path sits behind Keras `Model.predict / load_weights / save_weights`
 * on objects built in reference FRVSR.py:27-73 and TecoGAN.py:28-85 from the
 * blocks of reference layers.py:15-140.  Each entry point below names the
 * reference interface it replaces.  All tensors are NHWC float32 in host or
 * device memory exactly as Keras `predict` takes / returns them; every call
 * returns 0 on success or a negative sr_status, with text in sr_last_error().
 *
 * Threading: one sr_ctx per (thread, GPU); a ctx is not thread-safe.
 * Ownership: the caller owns every buffer it passes; no pointer is retained.
 * There is NO CPU fallback: without a HIP device every compute call fails.
 */
#ifndef SR_HIP_H
#define SR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sr_ctx sr_ctx;

enum sr_status {
    SR_OK = 0,
    SR_ERR_ARG = -1,      /* bad argument (shape, NULL, batch > max_batch ...)   */
    SR_ERR_SHAPE = -2,    /* h or w not a multiple of 8, scale != 4, ch != 3      */
    SR_ERR_WEIGHTS = -3,  /* weights missing / wrong order / wrong shape          */
    SR_ERR_HIP = -4,      /* a HIP runtime call failed (no device, OOM, ...)      */
    SR_ERR_STATE = -5     /* call sequence error (push before begin ...)          */
};

/* arithmetic of the convolutions */
enum sr_dtype {
    SR_FP32 = 0,    /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 activations        */
    SR_BF16 = 1,    /* bf16 MFMA, bf16 activations in HBM, fp32 accumulate / epilogue     */
    SR_BF16X3 = 2,  /* fp32 activations, operands split hi+lo bf16, 3 bf16 MFMAs/product  */
    SR_F16MX6 = 3   /* SR_BF16X3, except SRNet's 20 residual-block convs on the split-planar path: product = f16*f16 +
                     * two MX-scaled FP6 (E2M3, E8M0 scale per 32 channels) cross terms, activations stored f16 + FP6 residue
                     * (conv_trunk6.hip).  Same 1e-3 bar as the others; weights changed by sr_train_apply
                     * put the ctx back on the SR_BF16X3 kernels until the next sr_commit_weights */
};

enum sr_model { SR_FRVSR = 0, SR_TECOGAN = 1 };

/* activation fused into a convolution's epilogue (per-op entry points) */
enum sr_act { SR_ACT_NONE = 0, SR_ACT_RELU = 1, SR_ACT_LRELU02 = 2, SR_ACT_TANH = 3, SR_ACT_CLIP01 = 4 };

/* text of the last error on this ctx (ctx may be NULL: last error of a failed
 * sr_create or of a per-op call on this thread).  Never NULL. */
const char* sr_last_error(const sr_ctx* ctx);

/* Number of visible HIP devices (0 when there is none); never fails. */
int sr_device_count(void);

/* Replaces FrameRecurrentVideoSR.build / TecoGAN.build (reference FRVSR.py:27-66,
 * TecoGAN.py:28-80): fixes the LR tile shape (h, w, ch) and scale like
 * main.py:36-46 does, allocates every activation buffer for up to max_batch
 * tiles/clips.  h and w must be multiples of 8 (FNet's 3x pool / 3x upsample,
 * layers.py:31-41), ch == 3, scale == 4; otherwise SR_ERR_SHAPE. */
int sr_create(sr_ctx** out, int device, int h, int w, int ch, int scale,
              int max_batch, int dtype, int model);
/* sr_create for LR sizes that are not multiples of 8 (SURVEY.md 8(b) "unless padding mode requested"; the 180-row
 * frames of BASELINE's 180p -> 720p configs).  The reference itself cannot run such a size: FNet's pooled /
 * upsampled pyramid comes back 184 rows high and dense_image_warp (layers.py:131) would get mismatched shapes,
 * and main.py:36-41 only builds square tiles.  Here every tensor the caller passes or receives keeps its own
 * size (h, w) / (4h, 4w); inside, inputs are zero padded at the bottom / right to the next multiples of 8 and
 * outputs are cropped: the results are those of the reference graph built at the padded size on zero-padded inputs,
 * cropped.  The recurrent state of sr_clip_begin / push is kept at the padded size.  Training entry points
 * refuse such a ctx (SR_ERR_STATE). */
int sr_create_padded(sr_ctx** out, int device, int h, int w, int ch, int scale,
                     int max_batch, int dtype, int model);
int sr_destroy(sr_ctx* ctx);

/* Run the ctx's kernels on this hipStream_t (default: a stream the ctx owns).  A step forks half of
 * its batch onto a second, internal stream and joins it back with events before the call's last
 * operation on this stream, so the caller sees plain stream order (environment SR_NO_DUAL=1: one
 * stream only; SR_LANES=n: n lanes). */
int sr_set_stream(sr_ctx* ctx, void* hip_stream);
int sr_sync(sr_ctx* ctx);

/* Replaces Model.load_weights (reference FRVSR.py:88,212,384; TecoGAN.py:158,537,709).
 * Feed the 76 tensors of the Keras file in file order (28 under "FNet/", 48
 * under "SRNet/"; kernel then bias per layer).  Like Keras, tensors are matched
 * by position inside their group and checked by shape; numeric name suffixes
 * are not interpreted.  kernel: (3,3,Cin,Cout) HWIO, transposed conv:
 * (3,3,Cout,Cin), bias: (Cout,).  sr_commit_weights packs them into the MFMA
 * fragment order of the ctx's dtype and uploads them. */
int sr_load_tensor(sr_ctx* ctx, const char* keras_name, const float* host,
                   const int64_t* shape, int rank);
int sr_commit_weights(sr_ctx* ctx);
/* Replaces Model.save_weights (reference FRVSR.py:193): tensor `index` (0..75)
 * in file order.  name_out / host_out may be NULL to query name / shape only. */
int sr_get_tensor(const sr_ctx* ctx, int index, char* name_out, int name_cap,
                  float* host_out, int64_t* shape_out, int* rank_out);

/* Replaces frvsr.predict / generator.predict (reference FRVSR.py:293,454;
 * TecoGAN.py:618,779): [I_LR_t (B,h,w,3), I_LR_t_1 (B,h,w,3), I_est_t_1
 * (B,4h,4w,3)] -> [I_est_t (B,4h,4w,3), I_LR_est_t_1 (B,h,w,3)].  Stateless.
 * Host pointers; lr_warp may be NULL (the eval loops discard it). */
int sr_step(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1,
            const float* est_tm1, float* est_t, float* lr_warp);
/* Same with device pointers, asynchronous on the ctx stream (no host copies). */
int sr_step_dev(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1,
                const float* est_tm1, float* est_t, float* lr_warp);
/* Replaces hr_warping.predict (reference TecoGAN.py:83-85, call sites :329-333):
 * same three inputs -> I_hat_est_t_1 (B,4h,4w,3), the warped previous estimate. */
int sr_step_hr_warp(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1,
                    const float* est_tm1, float* hr_warp);

/* The recurrent driver of reference FRVSR.py:231-234,293,309 with the state kept
 * on the device: sr_clip_begin zeroes prev-LR and prev-estimate for B clips
 * (tiles) advanced in lock-step; each sr_clip_push consumes frame t of all B
 * clips (B,h,w,3) and, if est_t != NULL, copies the estimate (B,4h,4w,3) out.
 * _dev variants take device pointers and do not synchronise. */
int sr_clip_begin(sr_ctx* ctx, int B);
int sr_clip_push(sr_ctx* ctx, const float* lr_t, float* est_t);
int sr_clip_push_dev(sr_ctx* ctx, const float* lr_t, float* est_t);

/* The eval loop's output path (reference FRVSR.py:293-351, TecoGAN.py:618-676): predict, then
 * utils.denormalize = (x * 255).astype(uint8) (utils.py:61-63: truncation, no rounding) before the frame is
 * written out.  Like sr_clip_push, but the estimate leaves the device as uint8 (B,4h,4w,3) -- a quarter of
 * the bytes -- into est_u8_host (page-locked memory makes the copy asynchronous), converted on the ctx stream and
 * copied on a second stream so that the copy of frame t runs under the kernels of frame t+1.  lr_t is a host
 * pointer (lr_on_device = 0; page-locked for an asynchronous copy) or a device pointer.  The call returns once the
 * work is enqueued: est_u8_host is complete after sr_u8_wait(ctx, 0) (or sr_sync); sr_u8_wait(ctx, 1) returns
 * when every frame but the newest has landed -- the pattern is push(t) into buffer t % 2, then
 * sr_u8_wait(ctx, 1), then consume buffer (t-1) % 2.  Host LR frames are copied in on a third stream when the call is made,
 * i.e. under the kernels of the previous push.  The device serves copies in submission order, so the copy-out of frame t is
 * SUBMITTED by push t+1 (behind that push's copy-in) or by sr_u8_wait(ctx, 0) / sr_sync / sr_clip_begin: a frame that is
 * followed by neither never reaches the host.  lr_t must stay valid until the step that reads it has run (as before). */
int sr_clip_push_u8(sr_ctx* ctx, const float* lr_t, int lr_on_device, uint8_t* est_u8_host);
int sr_u8_wait(sr_ctx* ctx, int keep_in_flight);

/* The eval loop's metric frames (reference FRVSR.py:320-339: every 10th frame psnr / ssim read the float estimate
 * predict returned, not the uint8 frame written to the video): copies the CURRENT estimates -- the state the last
 * sr_clip_push* produced -- of clips first .. first + n - 1 to est (n,4h,4w,3) float32 on the host and returns when
 * they have landed.  For drivers that take the frames as uint8 (sr_clip_push_u8) and need the float values of a few. */
int sr_clip_read_est(sr_ctx* ctx, int first, int n, float* est);
/* The same read-back beside the NEXT step instead of in front of it (the eval loop's metric frames: 177 MB per 16 clips of 720p
 * would otherwise hold the device for ~7 ms).  _begin enqueues the copy of the newest estimates (clips [first, first + n)) into
 * `est` -- page-locked memory (sr_host_alloc), or the copy is not asynchronous -- on a stream of its own, ordered after the
 * step that produced them; the caller may push the next frame at once.  The push after that one waits on the device until the
 * copy is over (it overwrites the buffer being read).  _end blocks the host until `est` is valid.  One read-back at a time:
 * _begin before the previous _end is SR_ERR_STATE, as is _end without _begin.  The copy is submitted by the next push (behind
 * its copy-in, for the same reason as the uint8 frames') or by _end.  No counterpart in the reference (its
 * predict_on_batch returns every float frame to the host, FRVSR.py:292-296). */
int sr_clip_read_est_begin(sr_ctx* ctx, int first, int n, float* est);
int sr_clip_read_est_end(sr_ctx* ctx);

/* Page-locked host memory for the buffers sr_clip_push_u8 copies from / into asynchronously (Keras manages its own host
 * staging inside predict; no reference counterpart).  bytes > 0; sr_host_free(NULL) is a no-op. */
int sr_host_alloc(void** out, size_t bytes);
int sr_host_free(void* p);

/* Replaces fnet.predict (reference FRVSR.py:30, layers.py:15-49):
 * [I_LR_t, I_LR_t_1] (B,h,w,3) each -> flow (B,h,w,2). */
int sr_fnet(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1, float* flow);
/* Replaces srnet.predict (reference FRVSR.py:34, layers.py:51-80):
 * [I_LR_t (B,h,w,3), S_s (B,h,w,48)] -> (B,4h,4w,3); no TecoGAN epilogue. */
int sr_srnet(sr_ctx* ctx, int B, const float* lr_t, const float* s_s, float* out);

/* Per-kernel timing of the most recent steps: when enabled, every kernel launch
 * of a step is bracketed by HIP events on the ctx stream.  sr_profile_read
 * synchronises and returns, for layer i < *n, the accumulated milliseconds,
 * launch count, algorithmic FLOPs per launch and algorithmic HBM bytes per
 * launch (means over the launches that share the name), plus its name.  No reference counterpart (the reference times the whole
 * predict with datetime.now(), FRVSR.py:292-296). */
int sr_profile_enable(sr_ctx* ctx, int enable);
int sr_profile_count(const sr_ctx* ctx);
int sr_profile_read(sr_ctx* ctx, int i, char* name_out, int name_cap, double* ms_total,
                    int64_t* launches, double* flops_per_launch, double* bytes_per_launch);
int sr_profile_reset(sr_ctx* ctx);

/* ---- single ops (host pointers, NHWC float32), used by the parity tests and by
 * the weight-free sub-model facades.  `dtype` as enum sr_dtype. ---------------- */

/* Keras Conv2D(3x3, stride 1, padding='same') + activation (+ residual add after
 * the activation): reference layers.py:21-24,58-61.  w HWIO (3,3,Cin,Cout).
 * res may be NULL. */
int sr_op_conv3x3(int device, int dtype, int B, int H, int W, int Cin, int Cout,
                  const float* x, const float* w, const float* bias, int act,
                  const float* res, float* out);
/* Keras Conv2DTranspose(3x3, stride 2, padding='same'): reference layers.py:71-74.
 * w (3,3,Cout,Cin); out (B,2H,2W,Cout). */
int sr_op_convT3x3s2(int device, int dtype, int B, int H, int W, int Cin, int Cout,
                     const float* x, const float* w, const float* bias, int act, float* out);
/* The same layer for 64 -> 64 channels in SR_F16MX6 arithmetic (conv_trunk6.hip; reference layers.py:58-62, one conv of a residual
 * block): x, res, out (B,H,W,64).  A ReLU layer (relu = 1, res NULL) or a residual layer (relu = 0, res: out = res + conv(x)).
 * mode 0: the result as stored in the mx6 format; 1: residual layer writing split-planar records (the trunk's last layer);
 * 2 / 3: no conv -- x after a round trip through the format (the stored value / its FP6 image alone; w, bias may be NULL). */
int sr_op_conv3x3_mx6(int device, int B, int H, int W, const float* x, const float* w, const float* bias, int relu,
                      const float* res, int mode, float* out);
/* MaxPool2D(2x2, padding='same'): reference layers.py:31.  out (B,ceil(H/2),ceil(W/2),C). */
int sr_op_maxpool2(int device, int B, int H, int W, int C, const float* x, float* out);
/* UpSampling2D(size=s, interpolation='bilinear') = TF1 legacy resize: reference
 * layers.py:37,89.  out (B,sH,sW,C). */
int sr_op_resize_bilinear(int device, int B, int H, int W, int C, int s, const float* x, float* out);
/* tf.contrib.image.dense_image_warp: reference layers.py:131.  flow (B,H,W,2). */
int sr_op_warp(int device, int B, int H, int W, int C, const float* img, const float* flow, float* out);
/* tf.space_to_depth(x, bs): reference layers.py:109.  out (B,H/bs,W/bs,bs*bs*C). */
int sr_op_space_to_depth(int device, int B, int H, int W, int C, int bs, const float* x, float* out);

/* LR synthesis of the reference's train / eval drivers (FRVSR.py:117-119, 274-277): cv2.GaussianBlur(hr, (5,5), 0)
 * (taps [1,4,6,4,1]/16, BORDER_REFLECT_101) followed by cv2.resize to a quarter of the size with INTER_CUBIC
 * (weights [-3,19,19,-3]/32 on each block of 4), fused into one 8 x 8 gather.  hr (N,H,W,C), H and W multiples
 * of 4 -> lr (N,H/4,W/4,C).  Host pointers. */
int sr_op_blur_cubic_downscale(int device, int N, int H, int W, int C, const float* hr, float* lr);
/* The bicubic x4 image of the reference's adversarial batch and metrics (TecoGAN.py:313, FRVSR.py:304): cv2.resize(lr, 4x,
 * INTER_CUBIC), 4-tap cubic (a = -0.75) at the fixed fractions of an exact x4 enlargement, border taps clamped.
 * lr (N,h,w,C) -> out (N,4h,4w,C).  Host pointers. */
int sr_op_cubic_upscale4(int device, int N, int h, int w, int C, const float* lr, float* out);

/* ---- training step (SURVEY.md section 8 f, rank 1) ------------------------------------------------
 * Replaces frvsr.train_on_batch([lr_t, lr_tm1, est_tm1], [hr_t, lr_t]) (reference FRVSR.py:154-156)
 * on the model compiled at FRVSR.py:69-71: loss = 1 * mean((I_est_t - hr_t)^2) + 1 * mean((I_LR_est_t_1
 * - lr_t)^2) (Keras mean_squared_error = mean over every element), optimizer Adam(lr) with Keras
 * 2.2.4 defaults (beta_1 0.9, beta_2 0.999, epsilon 1e-7, lr_t = lr*sqrt(1-beta_2^t)/(1-beta_1^t)).
 * prev-LR and prev-estimate are constant inputs (no gradient through time, FRVSR.py:140-151); the
 * warps are differentiated with respect to the flow only.  ctx dtype must be fp32 or bf16x3.  With
 * model SR_TECOGAN the same call is generator.train_on_batch of TecoGAN's generator pre-training
 * (reference TecoGAN.py:112-113,228): identical losses on tanh(SRNet + bilinear x4 (lr_t)).  losses[3] = {total, sr_loss, flow_loss} (Keras train_on_batch order, FRVSR.py:161).
 *
 * sr_train_grads: forward + backward of one batch (host pointers; _dev: device pointers); the
 *   gradient of all 76 tensors lands in one flat fp32 device vector in file order (kernel then
 *   bias per layer, Keras layouts) and, if d_grads != NULL, is copied to that DEVICE buffer of
 *   sr_train_param_count() floats -- the buffer a data-parallel job all-reduces (RCCL) before
 * sr_train_apply: Adam step from d_grads * grad_scale (NULL: the ctx's own gradient vector) on the
 *   device master weights, then re-pack of every MFMA weight buffer on the device; the next
 *   sr_step / sr_clip_push and sr_get_tensor (save_weights) see the updated weights.
 * sr_train_on_batch = sr_train_grads + sr_train_apply with the Keras defaults (single GPU). */
int sr_train_param_count(const sr_ctx* ctx, int64_t* n);
int sr_train_grads(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1,
                   const float* hr_t, float* d_grads, float* losses);
int sr_train_grads_dev(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1,
                       const float* hr_t, float* d_grads, float* losses);
int sr_train_read_grads(sr_ctx* ctx, float* host_out);
/* ---- the data-parallel exchange on RCCL itself (SURVEY.md section 8 e; no counterpart in the reference, which trains on one
 * device).  librccl.so is opened at run time by the first of these calls (SR_ERR_HIP with the loader's text if it is absent).
 * sr_rccl_unique_id: rank 0 makes the 128-byte communicator id; the host program carries it to the other ranks (any channel;
 * dp.py uses its torch.distributed group).  sr_rccl_comm_create: every rank, collectively, with its device current; *comm is
 * an ncclComm_t.  sr_train_allreduce: ncclAllReduce(sum) in place on the flat gradient of the last sr_train_grads /
 * sr_train_clips_grads -- d_grads, or the ctx's own buffer when NULL -- enqueued on the ctx stream: sr_train_apply(grad_scale =
 * 1 / ranks) behind it needs no host synchronisation in between.  sr_rccl_allreduce_f32: the same for any device buffer and
 * stream (the adversarial step's discriminator gradient, loss values). */
int sr_rccl_unique_id(unsigned char* id128);
int sr_rccl_comm_create(void** comm, const unsigned char* id128, int rank, int nranks, int device);
int sr_rccl_comm_destroy(void* comm);
int sr_rccl_allreduce_f32(void* comm, float* d_buf, int64_t n, void* hip_stream);
int sr_train_allreduce(sr_ctx* ctx, void* comm, float* d_grads);
int sr_train_apply(sr_ctx* ctx, const float* d_grads, float grad_scale, float lr, float beta1,
                   float beta2, float eps);
int sr_train_on_batch(sr_ctx* ctx, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1,
                      const float* hr_t, float lr, float* losses);
int64_t sr_train_iterations(const sr_ctx* ctx);   /* optimizer steps taken (Keras `iterations`) */

/* One training iteration from HR clips, the loop body of reference FRVSR.py:103-156 (and TecoGAN.py:181-231)
 * without its host round trips: hr = HOST array [n_clips, c_frames, 4h, 4w, 3], uint8 (hr_is_u8 = 1, the
 * decoded frames; normalised on the device like utils.py:57-58) or float32 in [0,1].  On the device: LR =
 * GaussianBlur 5x5 + cubic x1/4 (FRVSR.py:117-119), previous LR frame (zeros at frame 0), previous estimate =
 * the generator's own no-gradient output for the previous frame, clips in lock step (zeros at frame 0), then
 * forward + backward as sr_train_grads.  n_clips * c_frames <= max_batch.  sr_train_clips_grads leaves the
 * flat gradient in the ctx (and in d_grads, a DEVICE pointer, when given) for sr_train_apply;
 * sr_train_on_clips applies Keras-default Adam with learning rate lr itself.  losses as sr_train_on_batch. */
int sr_train_clips_grads(sr_ctx* ctx, int n_clips, int c_frames, const void* hr, int hr_is_u8,
                         float* d_grads, float* losses);
int sr_train_on_clips(sr_ctx* ctx, int n_clips, int c_frames, const void* hr, int hr_is_u8, float lr,
                      float* losses);

/* Gradients of the single ops (host pointers, NHWC float32), for the parity tests against autograd
 * (dtype fp32 or bf16x3 selects the arithmetic of the convolution gradients):
 * Conv2D 3x3 'same' (x, w HWIO, dy) -> dx (nullable), dw (3,3,Cin,Cout), db (nullable);
 * Conv2DTranspose 3x3 s2 (x (B,H,W,Cin), w (3,3,Cout,Cin), dy (B,2H,2W,Cout)) -> dx, dw, db
 *   (dx needs Cout % 64 == 0, the layers of reference layers.py:71-74);
 * MaxPool2D 2x2 'same' (first maximum takes the gradient); legacy bilinear resize (adjoint);
 * dense_image_warp with respect to the flow. */
int sr_op_conv3x3_bwd(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* x,
                      const float* w, const float* dy, float* dx, float* dw, float* db);
int sr_op_convT3x3s2_bwd(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* x,
                         const float* w, const float* dy, float* dx, float* dw, float* db);
int sr_op_maxpool2_bwd(int device, int B, int H, int W, int C, const float* x, const float* gy, float* gx);
int sr_op_resize_bilinear_bwd(int device, int B, int H, int W, int C, int s, const float* gy, float* gx);
int sr_op_warp_bwd_flow(int device, int B, int H, int W, int C, const float* img, const float* flow,
                        const float* g, float* gflow);

/* ---- TecoGAN discriminator and adversarial step (SURVEY.md section 8 f, rank 3) -------------------------------
 * sr_disc replaces the Keras model of reference layers.py:142-186 (built at TecoGAN.py:60): nine HR images (B,H,W,3)
 * [hr_past, hr_pres, hr_fut, warp_past, warp_pres, warp_fut, bic_past, bic_pres, bic_fut] -> concat 27 -> Conv2D 3x3 64 +
 * LeakyReLU(0.2) -> 4 x [Conv2D 4x4 strides 2 'same' + BatchNormalization (epsilon 1e-3, momentum 0.99) + LeakyReLU(0.2)]
 * with 64 / 64 / 128 / 256 filters -> Flatten -> Dense(1) -> sigmoid; outputs the probability and the four feature maps.
 * H, W multiples of 16; dtype fp32 or bf16x3.  The 28 tensors in Discriminator.save_weights order: conv2d (kernel, bias),
 * 4 x [conv2d (kernel (4,4,Cin,Cout), bias), batch_normalization (gamma, beta, moving_mean, moving_variance)], dense
 * (kernel (H/16*W/16*256, 1), bias). */
typedef struct sr_disc sr_disc;
const char* sr_disc_last_error(const sr_disc* d);
int sr_disc_create(sr_disc** out, int device, int H, int W, int max_batch, int dtype);
int sr_disc_destroy(sr_disc* d);
int sr_disc_sync(sr_disc* d);
int sr_disc_tensor_count(void);
int sr_disc_param_count(const sr_disc* d, int64_t* n);
/* discriminator.load_weights / save_weights (reference TecoGAN.py:161,518): tensor `index` (0..27) in file order */
int sr_disc_set_tensor(sr_disc* d, int index, const float* host, const int64_t* shape, int rank);
int sr_disc_commit(sr_disc* d);
int sr_disc_get_tensor(sr_disc* d, int index, char* name_out, int name_cap, float* host_out, int64_t* shape_out, int* rank_out);
/* discriminator.predict (reference TecoGAN.py:446,467): BatchNormalization in inference mode (moving statistics).
 * imgs: 9 host pointers; prob (B); feats: 4 host pointers (B,H/2,W/2,64) ... (B,H/16,W/16,256); the array or entries may be NULL */
int sr_disc_predict(sr_disc* d, int B, const float* const* imgs, float* prob, float* const* feats);
/* discriminator.train_on_batch (reference TecoGAN.py:436-458) up to the optimizer: loss = binary_crossentropy(D_st, labels)
 * (Keras: probability clipped to [1e-7, 1-1e-7]; the feature-map outputs carry loss weight 0), BatchNormalization in
 * training mode, moving statistics updated; the flat gradient (28 tensors in file order, zeros for the moving statistics)
 * stays on the device.  sr_disc_read_grads copies it to the host or to a DEVICE buffer (out_on_device = 1: the all-reduce
 * buffer of a data-parallel job).  sr_disc_apply = Keras-2.2.4 Adam with iteration `t` (the value AFTER this step: the
 * reference shares one Adam instance between generator, discriminator and tecogan, TecoGAN.py:111-131, so the caller owns
 * the counter) from d_grads * grad_scale (DEVICE; NULL = own gradient), moment slot 0 (discriminator.train_on_batch) or
 * 1 (tecogan.train_on_batch); reset_moments zeroes the slot first (the reference re-compiles the discriminator every
 * iteration, TecoGAN.py:436, which gives its train function fresh Adam slots). */
int sr_disc_train_grads(sr_disc* d, int B, const float* const* imgs, const float* labels, float* loss);
int sr_disc_read_grads(sr_disc* d, float* out, int out_on_device);
int sr_disc_apply(sr_disc* d, const float* d_grads, float grad_scale, float lr, float beta1, float beta2, float eps, int64_t t,
                  int slot, int reset_moments);
/* tecogan.train_on_batch (reference TecoGAN.py:87-108,122-131,469-479) up to the optimizer, ctx model SR_TECOGAN:
 * inputs [lr_t, lr_tm1, est_tm1, next_est (I_hr_fut), next_hat_est (I_hr_warp_fut), bic_past, bic_pres, bic_fut], targets
 * [hr_t, ones, d_1..d_4 (d_targets: 4 host pointers), hr_t x 4 (VGG19 terms: evaluated when sr_tecogan_set_vgg attached the
 * weights, else 0 -- ImageNet weights are a download), pp_target (the batch-flipped hr_t), lr_t]; loss_weights: the 12 of TecoGAN.py:127-130 (the warp weight must
 * be 1); losses[13] = total + the 12 unweighted parts in Keras' order.  BatchNormalization runs in training mode (learning
 * phase 1) without touching the moving statistics.  Generator gradient -> the ctx (sr_train_read_grads), discriminator
 * gradient -> the discriminator (sr_disc_read_grads); sr_tecogan_read_grads_dev copies both to DEVICE buffers (NULL skips
 * one), the all-reduce buffers of a data-parallel job.  sr_tecogan_apply: Adam step `t` from d_gen_grads / d_disc_grads *
 * grad_scale (DEVICE; NULL = own gradient, unscaled) on the generator (moments of tecogan's
 * own train function) and, with train_disc != 0, on the discriminator: the reference compiles tecogan before it ever sets
 * discriminator.trainable = False (TecoGAN.py:122 vs :461), so Keras 2.2.4 keeps the discriminator's weights among
 * tecogan's collected trainable weights. */
int sr_tecogan_grads(sr_ctx* ctx, sr_disc* d, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1,
                     const float* next_est, const float* next_hat_est, const float* bic_past, const float* bic_pres,
                     const float* bic_fut, const float* hr_t, const float* const* d_targets, const float* pp_target,
                     const float* loss_weights, float* losses);
int sr_tecogan_apply(sr_ctx* ctx, sr_disc* d, const float* d_gen_grads, const float* d_disc_grads, float grad_scale, float lr,
                     float beta1, float beta2, float eps, int64_t t, int train_disc);
int sr_tecogan_read_grads_dev(sr_ctx* ctx, sr_disc* d, float* d_gen_out, float* d_disc_out);
/* One whole pass of the adversarial loop body (reference TecoGAN.py:434-479) on one GPU, every host array uploaded once:
 * arrays[15] = {lr, prev_lr, hr, prev_hr, next_hr, prev_hat_hr, next_hat_hr, est, prev_est, next_est, prev_hat_est,
 * next_hat_est, bic, prev_bic, next_bic} (the *_batch arrays of TecoGAN.py:416-432; LR (B,h,w,3), the rest (B,4h,4w,3)).
 * Runs discriminator.train_on_batch on the real triplets (labels 1; fresh Adam moments as after TecoGAN.py:436) and on
 * the generated ones (labels 0), discriminator.predict on the real triplets (the feature-map targets, kept on the
 * device), and tecogan.train_on_batch twice with the batch-flipped hr as ping-pong target (TecoGAN.py:469-479).  t0 = the
 * shared Adam counter before the pass (it advances by 4).  losses[13]: the last combined step; disc_losses[2] (nullable):
 * the two discriminator losses.  Same arithmetic as the step-by-step entry points (tests/test_gpu_disc.py). */
int sr_tecogan_iteration(sr_ctx* ctx, sr_disc* d, int B, const float* const* arrays, const float* loss_weights, float lr,
                         float beta1, float beta2, float eps, int64_t t0, int train_disc, float* losses, float* disc_losses);

/* TecoGAN's VGG19 perceptual terms (reference TecoGAN.py:125,129,867-901: vgg_22 / vgg_34 / vgg_44 / vgg_54 =
 * K.mean(K.square(f(y_true) - f(y_pred))) with f = keras.applications.VGG19(include_top=False) up to block2_conv2 /
 * block3_conv4 / block4_conv4 / block5_conv4, ReLU included, images fed as they are).  The reference downloads the ImageNet
 * weights; here the caller supplies the 32 tensors (index 2 l = kernel (3,3,Cin,Cout) HWIO, 2 l + 1 = bias of conv l in
 * Keras order block1_conv1 .. block5_conv4) -- without them the combined step reports the four terms as 0.  H, W: the HR
 * size, multiples of 16; dtype SR_FP32 or SR_BF16X3.  sr_vgg_loss_grads: host images (B,H,W,3) in, losses4 = the four
 * unweighted terms, grad_pred (nullable, (B,H,W,3)) = gradient of sum weights4[k] * loss_k with respect to y_pred (nothing
 * of VGG19 is trainable, TecoGAN.py:869-873).  sr_tecogan_set_vgg attaches the object to a SR_TECOGAN ctx (NULL detaches):
 * sr_tecogan_grads / sr_tecogan_iteration then evaluate outputs 6..9 on (hr_t, I_est_t) with loss_weights[6..9] and add
 * their gradient to the generator's.  The ctx does not own the object: detach before sr_vgg_destroy. */
typedef struct sr_vgg sr_vgg;
const char* sr_vgg_last_error(const sr_vgg* v);
int sr_vgg_tensor_count(void);
int sr_vgg_create(sr_vgg** out, int device, int H, int W, int max_batch, int dtype);
int sr_vgg_destroy(sr_vgg* v);
int sr_vgg_set_tensor(sr_vgg* v, int index, const float* host, const int64_t* shape, int rank);
int sr_vgg_commit(sr_vgg* v);
int sr_vgg_loss_grads(sr_vgg* v, int B, const float* y_true, const float* y_pred, const float* weights4, float* losses4,
                      float* grad_pred);
int sr_tecogan_set_vgg(sr_ctx* ctx, sr_vgg* v);

/* The ONE entry point that allocates nothing.  It has no counterpart in the reference; it exists for callers
 * that cannot call hipMalloc: code under HIP graph capture, tight training loops, harnesses that forbid
 * allocation.  It is sr_op_conv3x3_bwd's kernel / bias gradient (dw (3,3,Cin,Cout) HWIO, db (Cout,), db may be
 * NULL) of Conv2D 3x3 'same' from x (B,H,W,Cin) and dy (B,H,W,Cout), but on DEVICE pointers (dense NHWC
 * float32), asynchronous on `hip_stream` (a hipStream_t, NULL = the default stream), and with the per-slice
 * partial sums of its kernels in a static device array of SR_NOALLOC_SCRATCH_BYTES that every GPU the library
 * runs on holds until the process ends, instead of a hipMalloc'ed workspace.  Shapes whose partials do not fit
 * are rejected with SR_ERR_ARG: the partials take 37 KB x ceil(Cin/32) x ceil(Cout/32) x slice groups, and the
 * slice plan keeps that near 19 MB for any image size while Cin x Cout <= 2048 x 1024 or so (every layer of
 * this network fits at any batch; 2048 -> 2048 channels would need 152 MB and is refused).  Every call on a GPU shares that GPU's array: calls that overlap on one
 * GPU (different streams or threads) race; calls on one stream are safe back to back.  It calls no hipMalloc /
 * hipFree; the first call on a device only looks up the array's address (hipGetSymbolAddress) -- a lookup that
 * is not allowed inside a stream capture, so a caller that captures makes one warm-up call per device (any valid
 * shape) BEFORE it begins capturing.  `device` need not be the caller's current device: the call switches to it
 * and restores the caller's current device before it returns, on every path. */
#define SR_NOALLOC_SCRATCH_BYTES (32 * 1024 * 1024)
int sr_op_conv3x3_wgrad_noalloc(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* d_x,
                                const float* d_dy, float* d_dw, float* d_db, void* hip_stream);

/* Measurement utility (no reference counterpart): times `iters` back-to-back launches
 * of one 3x3 conv (or stride-2 transposed conv) layer on device-resident random data
 * with HIP events; used by tools/conv_bench.py to iterate on the MFMA kernels. */
int sr_bench_conv(int device, int dtype, int transposed, int B, int H, int W, int Cin, int Cout,
                  int residual, int iters, float* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* SR_HIP_H */
