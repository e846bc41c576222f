// Launchers of disc_kernels.hip (TecoGAN discriminator + adversarial losses), gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sr {

struct Ptr9 {
    const float* p[9];
};
// [npix,3] x 9 -> [npix,32] (channels 27..31 zero): Concatenate of reference layers.py:165-167
hipError_t launch_concat9(const Ptr9& p, float* out, long npix, hipStream_t s);
// space_to_depth(2) / depth_to_space(2) on NHWC float, C a multiple of 4: [B,2h,2w,C] <-> [B,h,w,4C], channel (dy*2+dx)*C + c
hipError_t launch_s2d2(const float* in, float* out, int B, int h, int w, int C, hipStream_t s);
hipError_t launch_d2s2(const float* in, float* out, int B, int h, int w, int C, hipStream_t s);

// BatchNormalization over the N rows of [N,C], C in {64,128,256}
size_t bn_workspace_bytes(long N, int C);
hipError_t launch_bn_stats(const float* z, long N, int C, double* partial, float eps, float momentum, float* mean, float* invstd,
                           float* moving_mean, float* moving_var, hipStream_t s);
hipError_t launch_bn_moving(const float* mm, const float* mv, int C, float eps, float* mean, float* invstd, hipStream_t s);
hipError_t launch_bn_apply(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                           long N, int C, hipStream_t s);
hipError_t launch_bn_bwd(const float* gy, const float* y, const float* z, const float* mean, const float* invstd, const float* gamma,
                         double* partial, float* dbeta, float* dgamma, float* dz, long N, int C, hipStream_t s);

hipError_t launch_dense_fwd(const float* y, const float* w, const float* bias, float* logit, int B, long F, hipStream_t s);
hipError_t launch_dense_bwd(const float* y, const float* w, const float* dlogit, float* dw, float* db, float* gy, int B, long F,
                            hipStream_t s);
hipError_t launch_bce(const float* logit, const float* labels, float label_const, int B, float scale, float* prob, float* dlogit,
                      float* loss_out, hipStream_t s);

size_t sqdiff_workspace_bytes(long n);
hipError_t launch_sqdiff_grad(const float* a, const float* t, float* g, double* partial, float* loss_out, long n, float weight,
                              int accumulate, hipStream_t s);
hipError_t launch_gather_w4(const float* dw3, float* dw4, int C, int O, hipStream_t s);

size_t teco_est_grad_workspace_bytes(long npix);
hipError_t launch_teco_est_grad(const float* est, const float* hr, const float* pp, const float* g27, const float* s_out, float* g_out,
                                double* partial, float* loss_gt, float* loss_pp, long npix, float w_gt, float w_pp, hipStream_t s,
                                const float* gvgg = nullptr);   // gvgg [npix,3] (nullable): gradient of the VGG19 terms w.r.t. est
hipError_t launch_feat_mse_grad(const float* ft, const float* fp, float* g, double* partial, float* loss_out, long n, float weight,
                                int accumulate, hipStream_t s);
hipError_t launch_add_hat_s2d(const float* g27, float* g_sin, int B, int H, int W, int gcs, hipStream_t s);

}  // namespace sr
