// TecoGAN's spatio-temporal discriminator (reference layers.py:142-186) on the GPU: state and internal entry points
// shared by sr_disc.hip (C ABI, forward / backward) and sr_train.hip (the combined generator + discriminator step).
#pragma once
#include "disc_kernels.h"
#include "sr_internal.h"

struct sr_disc {
    int device = 0, H = 0, W = 0, max_batch = 0, prec = sr::P_FP32;
    hipStream_t stream = nullptr;
    std::string err;
    // the 28 tensors of Discriminator.save_weights in file order: conv2d (kernel, bias), 4 x [conv2d (kernel, bias),
    // batch_normalization (gamma, beta, moving_mean, moving_variance)], dense (kernel, bias)
    std::vector<srh::HostTensor> tensors;
    bool committed = false, host_stale = false;
    int64_t off[29] = {0}, nparams = 0;
    float *d_params = nullptr, *d_grad = nullptr;
    float *d_m[2] = {nullptr, nullptr}, *d_v[2] = {nullptr, nullptr};   // Adam slots: 0 = discriminator.train_on_batch, 1 = tecogan.train_on_batch
    srh::ConvLayer conv[5], conv_dg[5];     // conv[1..4]: the 4x4 stride-2 layers as 3x3 layers over space_to_depth(2)
    // activations of one forward (kept for the backward) and gradient work buffers, sized for max_batch
    float *x27 = nullptr, *a0 = nullptr, *s[4] = {}, *z[4] = {}, *y[4] = {}, *gA = nullptr, *gB = nullptr, *gS = nullptr;
    float *bn_mean[4] = {}, *bn_invstd[4] = {};
    float *logit = nullptr, *prob = nullptr, *dlogit = nullptr, *labels = nullptr;
    float* d_loss = nullptr;               // [0] bce, [1..4] feature-map mse
    double* red = nullptr;                 // reduction partials (batch norm, squared differences)
    float *wg_part = nullptr, *dw3 = nullptr;
    float* in9[9] = {};                    // device copies of the nine input images (host entry points)
    float* featT[4] = {};                  // device copies of feature-map targets
    std::vector<float*> it_buf;            // sr_tecogan_iteration: the thirteen HR arrays of one pass (allocated on first use)
};

namespace srh {

constexpr int DISC_CIN[4] = {64, 64, 64, 128}, DISC_COUT[4] = {64, 64, 128, 256};
constexpr float DISC_BN_EPS = 1e-3f, DISC_BN_MOMENTUM = 0.99f;
inline int disc_kernel_index(int layer) { return layer == 0 ? 0 : (layer <= 4 ? 2 + 6 * (layer - 1) : 26); }   // conv0, conv1..4, dense

int dfail(sr_disc* d, int code, const char* fmt, ...);
// forward on nine DEVICE images [B,H,W,3]; training: batch statistics (and, with update_moving, the moving averages follow)
int disc_forward(sr_disc* d, int B, const sr::Ptr9& imgs, bool training, bool update_moving);
// backward from d->dlogit (set by launch_bce): feature-map targets (device, nullable) add w[k] * mse(y_k, target_k) -- their
// losses land in d->d_loss[1..4]; want_wgrads: parameter gradients into d->d_grad; want_dx: d->gS = gradient of the 27
// (32) input channels [B,H,W,32]
int disc_backward(sr_disc* d, int B, const float* const* feat_targets, const float* feat_w, bool want_wgrads, bool want_dx);
// Adam (Keras 2.2.4) on the trainable tensors with moment slot `slot`, then re-pack of the conv weights
int disc_apply(sr_disc* d, const float* d_grads, float grad_scale, float lr, float beta1, float beta2, float eps, int64_t t, int slot,
               bool reset_moments);

}  // namespace srh
