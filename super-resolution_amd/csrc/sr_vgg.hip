// TecoGAN's VGG19 perceptual terms (reference TecoGAN.py:125,129,867-901) behind the C ABI of include/sr_hip.h (sr_vgg_*).
//
// The reference builds keras.applications.VGG19(include_top=False, weights='imagenet') inside each of four loss functions
// and compares the block2_conv2 / block3_conv4 / block4_conv4 / block5_conv4 feature maps (post-ReLU: the activation is
// inside the Keras layer) of the ground truth and of the estimate, both fed as they are (BGR in [0, 1] / (-1, 1), no
// preprocess_input): K.mean(K.square(f(y_true) - f(y_pred))).  The ImageNet weights are a download; here the 32 tensors
// come from the caller (a file, `train.vgg19_weights`), in Keras order.  Nothing of VGG19 is trainable (TecoGAN.py:869-873):
// only the gradient with respect to the estimate is needed, so the pass is
//   forward(target), forward(estimate, kept), four feature MSEs, data gradients back to the image
// on the generator's own kernels: conv3x3_lds_kernel forward (bias + ReLU epilogue) and, with mirrored / transposed weights,
// as the data gradient with the ReLU mask of the tensor the gradient lands on folded into its epilogue (ConvArgs::res_mask);
// the 64 -> 3 gradient of block1_conv1 on the taps-in-N kernel; maxpool2 / maxpool2_bwd as in FNet's training step.  No CPU
// fallback.
#include "sr_vgg.h"

#include <cstdio>

using namespace sr;

namespace srh {

extern thread_local std::string g_err;

static int vfail(sr_vgg* v, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    g_err = tmp;
    if (v) v->err = tmp;
    return code;
}

#define VCHK(v, call)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return srh::vfail(v, SR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// conv l of the 16: block, channels; VGG_TAP[l] = index of the loss term read at its output, or -1
static const int VGG_CIN[16] = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512};
static const int VGG_COUT[16] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512};
static const int VGG_LEVEL[16] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};     // resolution = (H, W) >> level
static const int VGG_TAP[16] = {-1, -1, -1, 0, -1, -1, -1, 1, -1, -1, -1, 2, -1, -1, -1, 3};
static const char* const VGG_NAME[16] = {"block1_conv1", "block1_conv2", "block2_conv1", "block2_conv2", "block3_conv1", "block3_conv2",
                                         "block3_conv3", "block3_conv4", "block4_conv1", "block4_conv2", "block4_conv3", "block4_conv4",
                                         "block5_conv1", "block5_conv2", "block5_conv3", "block5_conv4"};

static bool first_of_block(int l) { return l > 0 && VGG_LEVEL[l] != VGG_LEVEL[l - 1]; }

static int valloc(sr_vgg* v, float** p, size_t elems) {
    if (hipMalloc((void**)p, elems * 4) != hipSuccess) return vfail(v, SR_ERR_HIP, "hipMalloc of %zu bytes failed", elems * 4);
    v->owned.push_back(*p);
    return SR_OK;
}

// forward of the 16 convs from a padded image in v->x16; keep: every layer's output stays in v->y[l] (the estimate's pass),
// else two ping-pong buffers carry the activations and only the four tapped maps are copied to v->ft (the target's pass)
static int vgg_forward(sr_vgg* v, int B, bool keep) {
    hipStream_t st = v->stream;
    const float* x = v->x16;
    for (int l = 0; l < 16; ++l) {
        const int h = v->H >> VGG_LEVEL[l], w = v->W >> VGG_LEVEL[l];
        if (first_of_block(l)) {    // MaxPooling2D((2, 2), strides 2, 'valid') of the previous block's last output (even sizes)
            float* p = keep ? v->p[VGG_LEVEL[l] - 1] : (x == v->tA ? v->tB : v->tA);
            hipError_t e = launch_maxpool2(x, p, 0, B, 2 * h, 2 * w, VGG_CIN[l], st);
            if (e != hipSuccess) return vfail(v, SR_ERR_HIP, "launch maxpool2: %s", hipGetErrorString(e));
            x = p;
        }
        float* out = keep ? v->y[l] : (x == v->tA ? v->tB : v->tA);
        if (!keep && VGG_TAP[l] >= 0) out = v->ft[VGG_TAP[l]];
        const int c = VGG_COUT[l];
        int rc = run_conv(nullptr, v->prec, st, v->fwd[l], B, {x, h, w, out, c, c, 1, nullptr, 0, A_RELU}, VGG_NAME[l]);
        if (rc) return vfail(v, rc, "%s", g_err.c_str());
        x = out;
    }
    return SR_OK;
}

// The four losses of (d_true, d_pred) [B,H,W,3] device tensors into v->d_loss[0..3] and, if d_g3 != nullptr, the gradient of
// sum_k w[k] * loss_k with respect to d_pred into d_g3 [B,H,W,3].  Everything on v->stream, nothing synchronised.
int vgg_loss_grads_dev(sr_vgg* v, int B, const float* d_true, const float* d_pred, const float* w4, float* d_g3) {
    hipStream_t st = v->stream;
    const long npix = (long)B * v->H * v->W;
    hipError_t e;
    if ((e = launch_pad_channels(d_true, v->x16, 0, npix, 3, 16, st)) != hipSuccess) return vfail(v, SR_ERR_HIP, "launch pad_channels: %s", hipGetErrorString(e));
    int rc = vgg_forward(v, B, false);
    if (rc) return rc;
    if ((e = launch_pad_channels(d_pred, v->x16, 0, npix, 3, 16, st)) != hipSuccess) return vfail(v, SR_ERR_HIP, "launch pad_channels: %s", hipGetErrorString(e));
    if ((rc = vgg_forward(v, B, true))) return rc;
    // backward: g = gradient w.r.t. the PRE-activation of conv l, at that layer's resolution
    float* g = nullptr;
    for (int l = 15; l >= 0; --l) {
        const int h = v->H >> VGG_LEVEL[l], w = v->W >> VGG_LEVEL[l];
        const long n = (long)B * h * w * VGG_COUT[l];
        if (VGG_TAP[l] >= 0) {
            const int k = VGG_TAP[l];
            float* dst = g ? g : v->gA;      // the deepest tap starts the gradient
            if ((e = launch_feat_mse_grad(v->ft[k], v->y[l], d_g3 ? dst : nullptr, v->red, v->d_loss + k, n, w4[k], g ? 1 : 0, st)) != hipSuccess)
                return vfail(v, SR_ERR_HIP, "launch feat_mse_grad: %s", hipGetErrorString(e));
            g = dst;
        }
        if (!d_g3) continue;                 // losses only
        float* nx = g == v->gA ? v->gB : v->gA;
        const int ci = VGG_CIN[l];
        if (l == 0) {                        // block1_conv1: 64 -> 3 on the taps-in-N kernel, no activation below it
            rc = run_conv(nullptr, v->prec, st, v->dg[0], B, {g, h, w, d_g3, 3, 3, 1, nullptr, 0, A_NONE}, "vgg.dgrad0");
            if (rc) return vfail(v, rc, "%s", g_err.c_str());
        } else if (first_of_block(l)) {      // the input is a pooled tensor: plain data gradient, then the pooling's adjoint + ReLU mask
            rc = run_conv(nullptr, v->prec, st, v->dg[l], B, {g, h, w, nx, ci, ci, 1, nullptr, 0, A_NONE}, "vgg.dgrad");
            if (rc) return vfail(v, rc, "%s", g_err.c_str());
            if ((e = launch_maxpool2_bwd(v->y[l - 1], nx, g, B, 2 * h, 2 * w, ci, A_RELU, st)) != hipSuccess)
                return vfail(v, SR_ERR_HIP, "launch maxpool2_bwd: %s", hipGetErrorString(e));
            // g (the buffer the gradient came in) now holds the gradient at the finer level; keep the roles
        } else {                             // same block: data gradient with the ReLU mask of y[l-1] in its epilogue
            ConvIO io = {g, h, w, nx, ci, ci, 1, v->y[l - 1], ci, A_RELU};
            io.res_mask = 1;
            rc = run_conv(nullptr, v->prec, st, v->dg[l], B, io, "vgg.dgrad");
            if (rc) return vfail(v, rc, "%s", g_err.c_str());
            g = nx;
        }
    }
    return SR_OK;
}

}  // namespace srh

using namespace srh;

extern "C" {

const char* sr_vgg_last_error(const sr_vgg* v) { return v ? v->err.c_str() : g_err.c_str(); }

int sr_vgg_tensor_count(void) { return 32; }

int sr_vgg_create(sr_vgg** out, int device, int H, int W, int max_batch, int dtype) {
    if (!out) return vfail(nullptr, SR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (H < 16 || W < 16 || H % 16 || W % 16) return vfail(nullptr, SR_ERR_SHAPE, "VGG19 feature terms: HR size %dx%d must be a multiple of 16 (four 2x2 poolings)", H, W);
    if (max_batch < 1) return vfail(nullptr, SR_ERR_ARG, "max_batch must be >= 1");
    if (dtype != SR_FP32 && dtype != SR_BF16X3) return vfail(nullptr, SR_ERR_ARG, "dtype must be fp32 or bf16x3 (fp32 activations)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return vfail(nullptr, SR_ERR_HIP, "no HIP device available; libsr_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return vfail(nullptr, SR_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    sr_vgg* v = new sr_vgg();
    v->device = device;
    v->H = H;
    v->W = W;
    v->max_batch = max_batch;
    v->prec = dtype;
    auto bail = [&](int rc) {
        g_err = v->err;
        sr_vgg_destroy(v);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&v->stream) != hipSuccess) return bail(vfail(v, SR_ERR_HIP, "stream creation failed"));
    v->own_stream = true;
    const size_t L0 = (size_t)max_batch * H * W;
    int rc = valloc(v, &v->x16, L0 * 16);
    for (int l = 0; l < 16 && !rc; ++l) rc = valloc(v, &v->y[l], (L0 >> (2 * VGG_LEVEL[l])) * VGG_COUT[l]);
    for (int k = 0; k < 4 && !rc; ++k) rc = valloc(v, &v->p[k], (L0 >> (2 * (k + 1))) * VGG_COUT[k == 0 ? 1 : (k == 1 ? 3 : (k == 2 ? 7 : 11))]);
    const int tap_layer[4] = {3, 7, 11, 15};
    for (int k = 0; k < 4 && !rc; ++k) rc = valloc(v, &v->ft[k], (L0 >> (2 * VGG_LEVEL[tap_layer[k]])) * VGG_COUT[tap_layer[k]]);
    // ping-pong buffers: the largest tensor is 64 channels at full size
    for (float** p : {&v->tA, &v->tB, &v->gA, &v->gB})
        if (!rc) rc = valloc(v, p, L0 * 64);
    if (!rc) rc = valloc(v, &v->g3, L0 * 3);
    if (!rc) rc = valloc(v, &v->d_in[0], L0 * 3);
    if (!rc) rc = valloc(v, &v->d_in[1], L0 * 3);
    if (!rc) rc = valloc(v, &v->d_loss, 4);
    if (!rc && hipMalloc((void**)&v->red, sqdiff_workspace_bytes((long)L0 * 64)) != hipSuccess) rc = vfail(v, SR_ERR_HIP, "hipMalloc failed");
    if (rc) return bail(rc);
    v->tensors.resize(32);
    *out = v;
    return SR_OK;
}

int sr_vgg_destroy(sr_vgg* v) {
    if (!v) return SR_OK;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    for (float* p : v->owned) (void)hipFree(p);
    if (v->red) (void)hipFree(v->red);
    for (auto& L : v->fwd) free_layer(L);
    for (auto& L : v->dg) free_layer(L);
    if (v->own_stream && v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
    return SR_OK;
}

int sr_vgg_set_tensor(sr_vgg* v, int index, const float* host, const int64_t* shape, int rank) {
    if (!v) return vfail(nullptr, SR_ERR_ARG, "vgg is NULL");
    if (index < 0 || index >= 32 || !host || !shape) return vfail(v, SR_ERR_ARG, "tensor index %d out of range or NULL pointer", index);
    const int l = index / 2;
    const bool kernel = !(index & 1);
    const int64_t want[4] = {3, 3, VGG_CIN[l], VGG_COUT[l]};
    bool ok = kernel ? rank == 4 : rank == 1;
    for (int i = 0; ok && i < rank; ++i) ok = shape[i] == (kernel ? want[i] : (int64_t)VGG_COUT[l]);
    if (!ok) return vfail(v, SR_ERR_WEIGHTS, "%s/%s: unexpected shape (VGG19 order: kernel (3,3,%d,%d), bias (%d,))", VGG_NAME[l], kernel ? "kernel:0" : "bias:0",
                          VGG_CIN[l], VGG_COUT[l], VGG_COUT[l]);
    HostTensor& t = v->tensors[index];
    t.name = std::string(VGG_NAME[l]) + (kernel ? "/kernel:0" : "/bias:0");
    t.shape.assign(shape, shape + rank);
    size_t n = 1;
    for (int i = 0; i < rank; ++i) n *= (size_t)shape[i];
    t.data.assign(host, host + n);
    v->committed = false;
    return SR_OK;
}

int sr_vgg_commit(sr_vgg* v) {
    if (!v) return vfail(nullptr, SR_ERR_ARG, "vgg is NULL");
    for (int i = 0; i < 32; ++i)
        if (v->tensors[i].data.empty()) return vfail(v, SR_ERR_WEIGHTS, "VGG19 tensor %d of 32 was not set", i);
    VCHK(v, hipSetDevice(v->device));
    for (int l = 0; l < 16; ++l) {
        plan_layer(v->fwd[l], v->prec, M_CONV, VGG_CIN[l], VGG_COUT[l]);
        int rc = upload_layer(nullptr, v->fwd[l], v->prec, v->tensors[2 * l].data.data(), v->tensors[2 * l + 1].data.data());
        if (rc) return vfail(v, rc, "%s", g_err.c_str());
        // data gradient: conv3x3 with the taps mirrored and the channel roles swapped (pack_entries PK_DGRAD), no bias
        plan_layer(v->dg[l], v->prec, M_CONV, VGG_COUT[l], VGG_CIN[l]);
        if ((rc = upload_layer(nullptr, v->dg[l], v->prec, v->tensors[2 * l].data.data(), nullptr, PK_DGRAD))) return vfail(v, rc, "%s", g_err.c_str());
    }
    v->committed = true;
    return SR_OK;
}

int sr_vgg_loss_grads(sr_vgg* v, int B, const float* y_true, const float* y_pred, const float* weights4, float* losses4, float* grad_pred) {
    if (!v) return vfail(nullptr, SR_ERR_ARG, "vgg is NULL");
    if (!v->committed) return vfail(v, SR_ERR_WEIGHTS, "VGG19 weights not committed (sr_vgg_set_tensor x32 + sr_vgg_commit)");
    if (B < 1 || B > v->max_batch) return vfail(v, SR_ERR_ARG, "batch %d outside [1, max_batch=%d]", B, v->max_batch);
    if (!y_true || !y_pred || !weights4) return vfail(v, SR_ERR_ARG, "NULL tensor pointer");
    VCHK(v, hipSetDevice(v->device));
    const size_t n = (size_t)B * v->H * v->W * 3;
    VCHK(v, hipMemcpyAsync(v->d_in[0], y_true, n * 4, hipMemcpyHostToDevice, v->stream));
    VCHK(v, hipMemcpyAsync(v->d_in[1], y_pred, n * 4, hipMemcpyHostToDevice, v->stream));
    int rc = vgg_loss_grads_dev(v, B, v->d_in[0], v->d_in[1], weights4, grad_pred ? v->g3 : nullptr);
    if (rc) return rc;
    float l4[4];
    VCHK(v, hipMemcpyAsync(l4, v->d_loss, sizeof l4, hipMemcpyDeviceToHost, v->stream));
    if (grad_pred) VCHK(v, hipMemcpyAsync(grad_pred, v->g3, n * 4, hipMemcpyDeviceToHost, v->stream));
    VCHK(v, hipStreamSynchronize(v->stream));
    if (losses4) memcpy(losses4, l4, sizeof l4);
    return SR_OK;
}

}  // extern "C"
