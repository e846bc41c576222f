// TecoGAN's spatio-temporal discriminator (reference layers.py:142-186) and its training step
// (reference TecoGAN.py:116-120,434-458) behind the C ABI of include/sr_hip.h (sr_disc_*).  No CPU fallback.
//
// Convolutions run on the generator's MFMA kernels.  Conv2D(4x4, strides 2, 'same') on an even size pads 1 / 1:
//     out[y, x] = sum_{ky,kx in 0..3} in[2y + ky - 1, 2x + kx - 1] W[ky, kx]
// Row 2y + ky - 1 is cell y + dy, phase py of the space_to_depth(2) image with ky = 2 dy + py + 1, dy in {-1, 0, 1}: the
// layer IS a 3x3 stride-1 'same' convolution over the space_to_depth image (4C channels, channel (py*2+px)*C + c) with
// the kernel W3[dy+1, dx+1, (py*2+px)*C + c, o] = W[ky, kx, c, o] (zero where ky or kx falls outside 0..3: 16 of the 36
// tap x phase cells are used).  Forward, data gradient (+ depth_to_space) and weight gradient (+ gather) all reuse
// conv3x3_lds_kernel / wgrad_kernel; the price is 2.25x the layer's MACs, bought back by not writing a second
// conv family for a model that is 1/3 of the adversarial step.
#include "sr_disc.h"

#include <cmath>
#include <cstdio>

using namespace sr;

namespace srh {

extern thread_local std::string g_err;

int dfail(sr_disc* d, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    g_err = tmp;
    if (d) d->err = tmp;
    return code;
}

#define DCHK(d, call)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return srh::dfail(d, SR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define DRC(call)                 \
    do {                          \
        int rc_ = (call);         \
        if (rc_) return rc_;      \
    } while (0)

namespace {

struct Spec {
    std::string name;
    std::vector<int64_t> shape;
};

std::vector<Spec> disc_specs(int H, int W) {
    std::vector<Spec> s;
    s.push_back({"conv2d_37/kernel:0", {3, 3, 27, 64}});
    s.push_back({"conv2d_37/bias:0", {64}});
    for (int k = 0; k < 4; ++k) {
        char nm[64];
        snprintf(nm, sizeof nm, "conv2d_%d/", 38 + k);
        s.push_back({std::string(nm) + "kernel:0", {4, 4, DISC_CIN[k], DISC_COUT[k]}});
        s.push_back({std::string(nm) + "bias:0", {DISC_COUT[k]}});
        snprintf(nm, sizeof nm, "batch_normalization_%d/", k + 1);
        for (const char* p : {"gamma:0", "beta:0", "moving_mean:0", "moving_variance:0"}) s.push_back({std::string(nm) + p, {DISC_COUT[k]}});
    }
    s.push_back({"dense_1/kernel:0", {(int64_t)(H / 16) * (W / 16) * 256, 1}});
    s.push_back({"dense_1/bias:0", {1}});
    return s;
}

// W [4,4,C,O] -> W3 [3,3,4C,O]
std::vector<float> expand_w4(const float* w, int C, int O) {
    std::vector<float> out((size_t)9 * 4 * C * O, 0.f);
    for (int ky = 0; ky < 4; ++ky)
        for (int kx = 0; kx < 4; ++kx) {
            const int ty = ky + 1, tx = kx + 1;
            const int tap = (ty >> 1) * 3 + (tx >> 1), ph = (ty & 1) * 2 + (tx & 1);
            for (int c = 0; c < C; ++c)
                memcpy(&out[((size_t)tap * 4 * C + (size_t)ph * C + c) * O], &w[((size_t)(ky * 4 + kx) * C + c) * O], (size_t)O * 4);
        }
    return out;
}

// entries that index W3 -> entries that index W (those on W3's structural zeros drop out)
void remap_ents_w4(std::vector<PackEnt>& ents, int C, int O) {
    std::vector<PackEnt> out;
    out.reserve(ents.size() / 2);
    for (const PackEnt& e : ents) {
        const int o = e.src % O;
        const int ci = (e.src / O) % (4 * C);
        const int tap = e.src / (O * 4 * C);
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, ph = ci / C, c = ci % C;
        const int ky = 2 * dy + (ph >> 1) + 1, kx = 2 * dx + (ph & 1) + 1;
        if (ky < 0 || ky > 3 || kx < 0 || kx > 3) continue;
        out.push_back({(int32_t)(((size_t)(ky * 4 + kx) * C + c) * O + o), e.dst});
    }
    ents.swap(out);
}

int upload_ents_d(sr_disc* d, ConvLayer& L) {
    if (L.d_ents) (void)hipFree(L.d_ents);
    L.d_ents = nullptr;
    if (L.ents.empty()) return SR_OK;
    DCHK(d, hipMalloc((void**)&L.d_ents, L.ents.size() * sizeof(PackEnt)));
    DCHK(d, hipMemcpy(L.d_ents, L.ents.data(), L.ents.size() * sizeof(PackEnt), hipMemcpyHostToDevice));
    return SR_OK;
}

int dalloc(sr_disc* d, float** p, size_t floats) {
    DCHK(d, hipMalloc((void**)p, (floats ? floats : 4) * sizeof(float)));
    return SR_OK;
}

// packed conv weights (forward + data gradient) and biases from the master parameters, on the device
int disc_repack(sr_disc* d) {
    hipStream_t st = d->stream;
    for (int l = 0; l < 5; ++l) {
        const int ki = disc_kernel_index(l);
        const float* k = d->d_params + d->off[ki];
        ConvLayer* Ls[2] = {&d->conv[l], &d->conv_dg[l]};
        for (ConvLayer* L : Ls) {
            if (!L->d_w) continue;
            if (hipError_t e = launch_repack(k, L->d_ents, (int)L->ents.size(), L->d_w, d->prec, st); e != hipSuccess)
                return dfail(d, SR_ERR_HIP, "launch repack: %s", hipGetErrorString(e));
        }
        DCHK(d, hipMemcpyAsync(d->conv[l].d_bias, d->d_params + d->off[ki + 1], (size_t)d->conv[l].cout * 4, hipMemcpyDeviceToDevice, st));
    }
    return SR_OK;
}

int ready(sr_disc* d, int B) {
    if (!d) return dfail(nullptr, SR_ERR_ARG, "discriminator is NULL");
    if (!d->committed) return dfail(d, SR_ERR_WEIGHTS, "weights not committed (sr_disc_set_tensor x28 + sr_disc_commit)");
    if (B < 1 || B > d->max_batch) return dfail(d, SR_ERR_ARG, "batch %d outside [1, max_batch=%d]", B, d->max_batch);
    DCHK(d, hipSetDevice(d->device));
    return SR_OK;
}

int stage_inputs(sr_disc* d, int B, const float* const* imgs, Ptr9& p) {
    const size_t n = (size_t)B * d->H * d->W * 3 * 4;
    for (int i = 0; i < 9; ++i) {
        if (!imgs || !imgs[i]) return dfail(d, SR_ERR_ARG, "input image %d is NULL", i);
        DCHK(d, hipMemcpyAsync(d->in9[i], imgs[i], n, hipMemcpyHostToDevice, d->stream));
        p.p[i] = d->in9[i];
    }
    return SR_OK;
}

}  // namespace

int disc_forward(sr_disc* d, int B, const Ptr9& imgs, bool training, bool update_moving) {
    hipStream_t st = d->stream;
    const int P = d->prec, H = d->H, W = d->W;
    if (hipError_t e = launch_concat9(imgs, d->x27, (long)B * H * W, st); e != hipSuccess) return dfail(d, SR_ERR_HIP, "concat9: %s", hipGetErrorString(e));
    DRC(run_conv(nullptr, P, st, d->conv[0], B, {d->x27, H, W, d->a0, 64, 64, 1, nullptr, 0, A_LRELU}, "disc.conv0"));
    const float* in = d->a0;
    int h = H, w = W;
    for (int k = 0; k < 4; ++k) {
        h /= 2;
        w /= 2;
        const int C = DISC_CIN[k], O = DISC_COUT[k];
        const int pi = 2 + 6 * k;
        DCHK(d, launch_s2d2(in, d->s[k], B, h, w, C, st));
        DRC(run_conv(nullptr, P, st, d->conv[1 + k], B, {d->s[k], h, w, d->z[k], O, O, 1, nullptr, 0, A_NONE}, "disc.conv4x4"));
        const long N = (long)B * h * w;
        if (training)
            DCHK(d, launch_bn_stats(d->z[k], N, O, d->red, DISC_BN_EPS, DISC_BN_MOMENTUM, d->bn_mean[k], d->bn_invstd[k],
                                    update_moving ? d->d_params + d->off[pi + 4] : nullptr, update_moving ? d->d_params + d->off[pi + 5] : nullptr, st));
        else
            DCHK(d, launch_bn_moving(d->d_params + d->off[pi + 4], d->d_params + d->off[pi + 5], O, DISC_BN_EPS, d->bn_mean[k], d->bn_invstd[k], st));
        DCHK(d, launch_bn_apply(d->z[k], d->bn_mean[k], d->bn_invstd[k], d->d_params + d->off[pi + 2], d->d_params + d->off[pi + 3], d->y[k], N, O, st));
        in = d->y[k];
    }
    if (training && update_moving) d->host_stale = true;
    const long F = (long)h * w * 256;
    DCHK(d, launch_dense_fwd(d->y[3], d->d_params + d->off[26], d->d_params + d->off[27], d->logit, B, F, st));
    return SR_OK;
}

int disc_backward(sr_disc* d, int B, const float* const* feat_targets, const float* feat_w, bool want_wgrads, bool want_dx) {
    hipStream_t st = d->stream;
    const int P = d->prec, H = d->H, W = d->W;
    int hs[4], ws[4];
    for (int k = 0; k < 4; ++k) {
        hs[k] = H >> (k + 1);
        ws[k] = W >> (k + 1);
    }
    const long F = (long)hs[3] * ws[3] * 256;
    float *cur = d->gA, *oth = d->gB;
    DCHK(d, launch_dense_bwd(d->y[3], d->d_params + d->off[26], d->dlogit, want_wgrads ? d->d_grad + d->off[26] : nullptr,
                             want_wgrads ? d->d_grad + d->off[27] : nullptr, cur, B, F, st));
    for (int k = 3; k >= 0; --k) {
        const int C = DISC_CIN[k], O = DISC_COUT[k], h = hs[k], w = ws[k];
        const int pi = 2 + 6 * k;
        const long N = (long)B * h * w;
        if (feat_targets && feat_targets[k])   // + w_k * d mse(y_k, target_k) / d y_k
            DCHK(d, launch_sqdiff_grad(d->y[k], feat_targets[k], cur, d->red, d->d_loss + 1 + k, N * O, feat_w[k], 1, st));
        // LeakyReLU and batch normalisation (batch statistics): dgamma, dbeta, then dz in place
        DCHK(d, launch_bn_bwd(cur, d->y[k], d->z[k], d->bn_mean[k], d->bn_invstd[k], d->d_params + d->off[pi + 2], d->red,
                              d->d_grad + d->off[pi + 3], d->d_grad + d->off[pi + 2], cur, N, O, st));
        if (want_wgrads) {
            WgradArgs wa = {d->s[k], cur, d->wg_part, d->dw3, d->d_grad + d->off[pi + 1], B, h, w, 4 * C, O, 4 * C, O, 0, P == P_BF16X3 ? 1 : 0, 0};
            DCHK(d, launch_wgrad(wa, st));
            DCHK(d, launch_gather_w4(d->dw3, d->d_grad + d->off[pi], C, O, st));
        }
        // data gradient in space_to_depth space, then back to the layer's input grid
        DRC(run_conv(nullptr, P, st, d->conv_dg[1 + k], B, {cur, h, w, d->gS, 4 * C, 4 * C, 1, nullptr, 0, A_NONE}, "disc.dgrad4x4"));
        DCHK(d, launch_d2s2(d->gS, oth, B, h, w, C, st));
        std::swap(cur, oth);
    }
    // cur = gradient of a0 = LeakyReLU(conv0)
    const long npix = (long)B * H * W;
    DCHK(d, launch_act_bwd(cur, d->a0, A_LRELU, npix, 64, 64, 64, st));
    if (want_wgrads) {
        WgradArgs wa = {d->x27, cur, d->wg_part, d->d_grad + d->off[0], d->d_grad + d->off[1], B, H, W, 27, 64, 32, 64, 0, P == P_BF16X3 ? 1 : 0, 0};
        DCHK(d, launch_wgrad(wa, st));
    }
    if (want_dx) DRC(run_conv(nullptr, P, st, d->conv_dg[0], B, {cur, H, W, d->gS, 32, 27, 1, nullptr, 0, A_NONE}, "disc.dgrad0"));
    return SR_OK;
}

int disc_apply(sr_disc* d, const float* d_grads, float grad_scale, float lr, float beta1, float beta2, float eps, int64_t t, int slot,
               bool reset_moments) {
    if (slot < 0 || slot > 1) return dfail(d, SR_ERR_ARG, "Adam slot must be 0 (discriminator) or 1 (tecogan)");
    if (t < 1) return dfail(d, SR_ERR_ARG, "Adam iteration must be >= 1");
    hipStream_t st = d->stream;
    const size_t nb = (size_t)d->nparams * 4;
    if (reset_moments) {
        DCHK(d, hipMemsetAsync(d->d_m[slot], 0, nb, st));
        DCHK(d, hipMemsetAsync(d->d_v[slot], 0, nb, st));
    }
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    // the gradient entries of the moving statistics are zero and never written: Adam leaves them alone
    DCHK(d, launch_adam(d->d_params, d->d_m[slot], d->d_v[slot], d_grads ? d_grads : d->d_grad, d->nparams, (float)lr_t, beta1, beta2, eps,
                        grad_scale, st));
    d->host_stale = true;
    return disc_repack(d);
}

}  // namespace srh

using namespace srh;

// =============================================================== C ABI
extern "C" {

const char* sr_disc_last_error(const sr_disc* d) { return d ? d->err.c_str() : g_err.c_str(); }
int sr_disc_tensor_count(void) { return 28; }

int sr_disc_destroy(sr_disc* d) {
    if (!d) return SR_OK;
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    float* fl[] = {d->d_params, d->d_grad, d->d_m[0], d->d_m[1], d->d_v[0], d->d_v[1], d->x27, d->a0, d->gA, d->gB, d->gS,
                   d->logit, d->prob, d->dlogit, d->labels, d->d_loss, d->wg_part, d->dw3};
    for (float* p : fl)
        if (p) (void)hipFree(p);
    for (int k = 0; k < 4; ++k)
        for (float* p : {d->s[k], d->z[k], d->y[k], d->bn_mean[k], d->bn_invstd[k], d->featT[k]})
            if (p) (void)hipFree(p);
    for (float* p : d->in9)
        if (p) (void)hipFree(p);
    for (float* p : d->it_buf)
        if (p) (void)hipFree(p);
    if (d->red) (void)hipFree(d->red);
    for (int l = 0; l < 5; ++l) {
        free_layer(d->conv[l]);
        free_layer(d->conv_dg[l]);
    }
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    return SR_OK;
}

int sr_disc_create(sr_disc** out, int device, int H, int W, int max_batch, int dtype) {
    if (!out) return dfail(nullptr, SR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (H < 16 || W < 16 || H % 16 || W % 16)
        return dfail(nullptr, SR_ERR_SHAPE, "discriminator input %dx%d: height and width must be multiples of 16 (four stride-2 convolutions)", H, W);
    if (max_batch < 1) return dfail(nullptr, SR_ERR_ARG, "max_batch must be >= 1");
    if (dtype != SR_FP32 && dtype != SR_BF16X3) return dfail(nullptr, SR_ERR_ARG, "discriminator dtype must be fp32 or bf16x3 (fp32 activations)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return dfail(nullptr, SR_ERR_HIP, "no HIP device available; libsr_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return dfail(nullptr, SR_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    sr_disc* d = new sr_disc();
    d->device = device;
    d->H = H;
    d->W = W;
    d->max_batch = max_batch;
    d->prec = dtype;
    auto bail = [&](int code) {
        g_err = d->err;
        sr_disc_destroy(d);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&d->stream) != hipSuccess) return bail(dfail(d, SR_ERR_HIP, "stream creation failed"));
    const std::vector<Spec> specs = disc_specs(H, W);
    for (size_t i = 0; i < specs.size(); ++i) {
        size_t n = 1;
        for (int64_t v : specs[i].shape) n *= (size_t)v;
        d->off[i + 1] = d->off[i] + (int64_t)n;
    }
    d->nparams = d->off[28];
    const size_t B = max_batch, L0 = (size_t)H * W;
    int rc = SR_OK;
    auto A = [&](float** p, size_t floats) { return rc ? rc : (rc = dalloc(d, p, floats)); };
    for (float** p : {&d->d_params, &d->d_grad, &d->d_m[0], &d->d_m[1], &d->d_v[0], &d->d_v[1]}) A(p, (size_t)d->nparams);
    A(&d->x27, B * L0 * 32);
    A(&d->a0, B * L0 * 64);
    A(&d->gA, B * L0 * 64);
    A(&d->gB, B * L0 * 64);
    A(&d->gS, B * L0 * 64);
    size_t red_bytes = sqdiff_workspace_bytes((long)(B * L0 * 64)), wg_bytes = wgrad_workspace_bytes((int)B, H, W, 27, 64, 0), dw3 = 0;
    for (int k = 0; k < 4; ++k) {
        const size_t px = L0 >> (2 * (k + 1));
        A(&d->s[k], B * px * 4 * DISC_CIN[k]);
        A(&d->z[k], B * px * DISC_COUT[k]);
        A(&d->y[k], B * px * DISC_COUT[k]);
        A(&d->featT[k], B * px * DISC_COUT[k]);
        A(&d->bn_mean[k], DISC_COUT[k]);
        A(&d->bn_invstd[k], DISC_COUT[k]);
        red_bytes = std::max(red_bytes, bn_workspace_bytes((long)(B * px), DISC_COUT[k]));
        wg_bytes = std::max(wg_bytes, wgrad_workspace_bytes((int)B, H >> (k + 1), W >> (k + 1), 4 * DISC_CIN[k], DISC_COUT[k], 0));
        dw3 = std::max(dw3, (size_t)9 * 4 * DISC_CIN[k] * DISC_COUT[k]);
    }
    for (float*& p : d->in9) A(&p, B * L0 * 3);
    A(&d->logit, B);
    A(&d->prob, B);
    A(&d->dlogit, B);
    A(&d->labels, B);
    A(&d->d_loss, 8);
    A(&d->wg_part, wg_bytes / 4 + 64);
    A(&d->dw3, dw3);
    if (rc) return bail(rc);
    if (hipMalloc((void**)&d->red, red_bytes + 256) != hipSuccess) return bail(dfail(d, SR_ERR_HIP, "hipMalloc failed"));
    for (float* p : {d->d_grad, d->d_m[0], d->d_m[1], d->d_v[0], d->d_v[1]})
        if (hipMemset(p, 0, (size_t)d->nparams * 4) != hipSuccess) return bail(dfail(d, SR_ERR_HIP, "hipMemset failed"));
    if (hipMemset(d->d_loss, 0, 32) != hipSuccess) return bail(dfail(d, SR_ERR_HIP, "hipMemset failed"));
    d->tensors.resize(28);
    for (int i = 0; i < 28; ++i) {
        d->tensors[i].name = specs[i].name;
        d->tensors[i].shape = specs[i].shape;
    }
    *out = d;
    return SR_OK;
}

// Replaces Model.load_weights on the discriminator (reference TecoGAN.py:161): tensor `index` in file order, checked by shape
int sr_disc_set_tensor(sr_disc* d, int index, const float* host, const int64_t* shape, int rank) {
    if (!d || !host || !shape) return dfail(d, SR_ERR_ARG, "NULL argument");
    if (index < 0 || index >= 28) return dfail(d, SR_ERR_ARG, "tensor index %d out of range [0, 28)", index);
    HostTensor& t = d->tensors[index];
    if (rank != (int)t.shape.size() || memcmp(shape, t.shape.data(), sizeof(int64_t) * rank))
        return dfail(d, SR_ERR_WEIGHTS, "%s: shape does not match position %d of the discriminator's weight list", t.name.c_str(), index);
    size_t n = 1;
    for (int i = 0; i < rank; ++i) n *= (size_t)shape[i];
    t.data.assign(host, host + n);
    d->committed = false;
    return SR_OK;
}

int sr_disc_commit(sr_disc* d) {
    if (!d) return dfail(nullptr, SR_ERR_ARG, "discriminator is NULL");
    for (int i = 0; i < 28; ++i)
        if (d->tensors[i].data.empty()) return dfail(d, SR_ERR_WEIGHTS, "%s (tensor %d) was not set", d->tensors[i].name.c_str(), i);
    DCHK(d, hipSetDevice(d->device));
    DCHK(d, hipStreamSynchronize(d->stream));
    for (int i = 0; i < 28; ++i)
        DCHK(d, hipMemcpy(d->d_params + d->off[i], d->tensors[i].data.data(), d->tensors[i].data.size() * 4, hipMemcpyHostToDevice));
    for (int l = 0; l < 5; ++l) {
        const int ki = disc_kernel_index(l);
        const float* w = d->tensors[ki].data.data();
        const float* b = d->tensors[ki + 1].data.data();
        std::vector<float> w3;
        int cin = 27, cout = 64;
        if (l > 0) {
            w3 = expand_w4(w, DISC_CIN[l - 1], DISC_COUT[l - 1]);
            w = w3.data();
            cin = 4 * DISC_CIN[l - 1];
            cout = DISC_COUT[l - 1];
        }
        free_layer(d->conv[l]);
        free_layer(d->conv_dg[l]);
        plan_layer(d->conv[l], d->prec, M_CONV, cin, cout);
        DRC(upload_layer(nullptr, d->conv[l], d->prec, w, b, PK_FWD));
        plan_layer(d->conv_dg[l], d->prec, M_CONV, cout, cin);
        DRC(upload_layer(nullptr, d->conv_dg[l], d->prec, w, nullptr, PK_DGRAD));
        if (l > 0) {
            remap_ents_w4(d->conv[l].ents, DISC_CIN[l - 1], DISC_COUT[l - 1]);
            remap_ents_w4(d->conv_dg[l].ents, DISC_CIN[l - 1], DISC_COUT[l - 1]);
        }
        DRC(upload_ents_d(d, d->conv[l]));
        DRC(upload_ents_d(d, d->conv_dg[l]));
    }
    d->committed = true;
    d->host_stale = false;
    return SR_OK;
}

int sr_disc_get_tensor(sr_disc* d, int index, char* name_out, int name_cap, float* host_out, int64_t* shape_out, int* rank_out) {
    if (!d) return dfail(nullptr, SR_ERR_ARG, "discriminator is NULL");
    if (index < 0 || index >= 28) return dfail(nullptr, SR_ERR_ARG, "tensor index %d out of range", index);
    HostTensor& t = d->tensors[index];
    if (d->host_stale && d->committed) {
        DCHK(d, hipSetDevice(d->device));
        DCHK(d, hipStreamSynchronize(d->stream));
        for (int i = 0; i < 28; ++i)
            DCHK(d, hipMemcpy(d->tensors[i].data.data(), d->d_params + d->off[i], d->tensors[i].data.size() * 4, hipMemcpyDeviceToHost));
        d->host_stale = false;
    }
    if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", t.name.c_str());
    if (rank_out) *rank_out = (int)t.shape.size();
    if (shape_out) memcpy(shape_out, t.shape.data(), t.shape.size() * sizeof(int64_t));
    if (host_out) {
        if (t.data.empty()) return dfail(d, SR_ERR_WEIGHTS, "%s was never set", t.name.c_str());
        memcpy(host_out, t.data.data(), t.data.size() * 4);
    }
    return SR_OK;
}

// Replaces discriminator.predict (reference TecoGAN.py:446,467): inference-mode BatchNormalization (moving statistics).
// imgs: nine host images [B,H,W,3] in the order of layers.py:165-167; prob [B]; feats: four host buffers
// [B,H/2,W/2,64], [B,H/4,W/4,64], [B,H/8,W/8,128], [B,H/16,W/16,256] (the array or any entry may be NULL)
int sr_disc_predict(sr_disc* d, int B, const float* const* imgs, float* prob, float* const* feats) {
    DRC(ready(d, B));
    Ptr9 p;
    DRC(stage_inputs(d, B, imgs, p));
    DRC(disc_forward(d, B, p, false, false));
    hipStream_t st = d->stream;
    DCHK(d, launch_bce(d->logit, nullptr, 0.f, B, 0.f, d->prob, nullptr, nullptr, st));
    if (prob) DCHK(d, hipMemcpyAsync(prob, d->prob, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    for (int k = 0; k < 4 && feats; ++k)
        if (feats[k])
            DCHK(d, hipMemcpyAsync(feats[k], d->y[k], (size_t)B * (d->H >> (k + 1)) * (d->W >> (k + 1)) * DISC_COUT[k] * 4, hipMemcpyDeviceToHost, st));
    DCHK(d, hipStreamSynchronize(st));
    return SR_OK;
}

// Forward (training-mode BatchNormalization, moving statistics updated) + backward of discriminator.train_on_batch
// (reference TecoGAN.py:436-450): loss = binary_crossentropy(D_st, labels); the four feature-map outputs carry loss weight 0.
// The gradient of the 28 tensors (zeros for the moving statistics) stays on the device for sr_disc_apply.
int sr_disc_train_grads(sr_disc* d, int B, const float* const* imgs, const float* labels, float* loss) {
    DRC(ready(d, B));
    if (!labels) return dfail(d, SR_ERR_ARG, "labels is NULL");
    Ptr9 p;
    DRC(stage_inputs(d, B, imgs, p));
    hipStream_t st = d->stream;
    DCHK(d, hipMemcpyAsync(d->labels, labels, (size_t)B * 4, hipMemcpyHostToDevice, st));
    DRC(disc_forward(d, B, p, true, true));
    DCHK(d, launch_bce(d->logit, d->labels, 0.f, B, 1.f, d->prob, d->dlogit, d->d_loss, st));
    DRC(disc_backward(d, B, nullptr, nullptr, true, false));
    float l = 0.f;
    DCHK(d, hipMemcpyAsync(&l, d->d_loss, 4, hipMemcpyDeviceToHost, st));
    DCHK(d, hipStreamSynchronize(st));
    if (loss) *loss = l;
    return SR_OK;
}

int sr_disc_param_count(const sr_disc* d, int64_t* n) {
    if (!d || !n) return dfail(nullptr, SR_ERR_ARG, "NULL argument");
    *n = d->nparams;
    return SR_OK;
}

// flat gradient of the last sr_disc_train_grads / sr_tecogan_grads (file order, Keras layouts) to the host, or to a
// DEVICE buffer (the all-reduce buffer of a data-parallel job)
int sr_disc_read_grads(sr_disc* d, float* out, int out_on_device) {
    if (!d || !out) return dfail(d, SR_ERR_ARG, "NULL argument");
    DCHK(d, hipSetDevice(d->device));
    DCHK(d, hipMemcpyAsync(out, d->d_grad, (size_t)d->nparams * 4, out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, d->stream));
    DCHK(d, hipStreamSynchronize(d->stream));
    return SR_OK;
}

// Keras-2.2.4 Adam step on the discriminator: d_grads (DEVICE, nparams floats) or NULL for its own gradient; `t` is the
// optimizer's iteration AFTER this step (the reference shares ONE Adam between generator, discriminator and tecogan,
// TecoGAN.py:111-131, so the counter lives with the caller); slot 0 / 1 selects the moments of discriminator.train_on_batch /
// tecogan.train_on_batch; reset_moments zeroes them first (the reference re-compiles the discriminator every iteration,
// TecoGAN.py:436, which gives its train function fresh Adam slots)
int sr_disc_apply(sr_disc* d, const float* d_grads, float grad_scale, float lr, float beta1, float beta2, float eps, int64_t t, int slot,
                  int reset_moments) {
    if (!d) return dfail(nullptr, SR_ERR_ARG, "discriminator is NULL");
    if (!d->committed) return dfail(d, SR_ERR_WEIGHTS, "weights not committed");
    DCHK(d, hipSetDevice(d->device));
    return disc_apply(d, d_grads, grad_scale, lr, beta1, beta2, eps, t, slot, reset_moments != 0);
}

int sr_disc_sync(sr_disc* d) {
    if (!d) return dfail(nullptr, SR_ERR_ARG, "discriminator is NULL");
    DCHK(d, hipSetDevice(d->device));
    DCHK(d, hipStreamSynchronize(d->stream));
    return SR_OK;
}

}  // extern "C"
