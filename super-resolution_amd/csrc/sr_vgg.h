// TecoGAN's VGG19 perceptual terms (reference TecoGAN.py:125,129,867-901): state and the internal entry point shared by
// sr_vgg.hip (C ABI) and sr_train.hip (the combined generator + discriminator step).
#pragma once
#include "disc_kernels.h"
#include "sr_internal.h"

struct sr_vgg {
    int device = 0, H = 0, W = 0, max_batch = 0, prec = sr::P_FP32;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::vector<srh::HostTensor> tensors;      // 32: block1_conv1/kernel:0, block1_conv1/bias:0, ... block5_conv4/bias:0 (Keras order)
    bool committed = false;
    srh::ConvLayer fwd[16], dg[16];            // forward convs; their data-gradient convs (mirrored / transposed packing)
    float* x16 = nullptr;                      // the input image padded to 16 channels
    float* y[16] = {};                         // post-ReLU outputs of the estimate's pass (kept for the backward)
    float* p[4] = {};                          // pooled tensors of the estimate's pass
    float* ft[4] = {};                         // the target's four tapped feature maps
    float *tA = nullptr, *tB = nullptr;        // activations of the target's pass (ping-pong)
    float *gA = nullptr, *gB = nullptr;        // gradients (ping-pong)
    float* g3 = nullptr;                       // gradient w.r.t. the estimate [B,H,W,3]
    float* d_in[2] = {nullptr, nullptr};       // device copies of the host entry point's images
    float* d_loss = nullptr;                   // [4]
    double* red = nullptr;
    std::vector<float*> owned;
};

namespace srh {
// losses of (d_true, d_pred) [B,H,W,3] -> v->d_loss[0..3]; d_g3 (nullable): gradient of sum w4[k] loss_k w.r.t. d_pred.
// Runs on v->stream, synchronises nothing.
int vgg_loss_grads_dev(sr_vgg* v, int B, const float* d_true, const float* d_pred, const float* w4, float* d_g3);
}  // namespace srh
