// Training step of the FRVSR / TecoGAN generator on the device: replaces frvsr.train_on_batch (reference
// FRVSR.py:156) for the model compiled at FRVSR.py:69-71 (two mean-squared-error losses with weights
// 1, 1 and Adam(lr=1e-4)) and generator.train_on_batch of TecoGAN's generator pre-training
// (TecoGAN.py:112-113,228: the same two losses on the generator with the tanh epilogue).  Forward with every activation kept, backward in reverse graph order,
// gradients into one flat fp32 vector in the Keras file order (the buffer a data-parallel job
// all-reduces), Adam on the flat master weights, device re-pack of the MFMA weight buffers.
//
//   forward / data gradients of the stride-1 convs: the inference MFMA kernels (conv_mfma.hip,
//   conv_tapsn.hip); the data gradient is the same convolution with mirrored, transposed weights
//   weight gradients, transposed-conv data gradient, gathers: train_kernels.hip
#include "sr_internal.h"

#include <condition_variable>
#include <mutex>
#include <thread>
#include "sr_disc.h"
#include "sr_vgg.h"

#include <cmath>
#include <cstdio>

using namespace sr;
using namespace srh;

struct sr_train_state {
    int64_t nparams = 0;
    int64_t off[77] = {0};              // offsets of the 76 tensors in the flat vectors
    float *d_params = nullptr, *d_m = nullptr, *d_v = nullptr, *d_grad = nullptr;
    int64_t t = 0;                      // optimizer iterations
    float *d_m2 = nullptr, *d_v2 = nullptr;   // Adam moments of the generator inside tecogan.train_on_batch (its own train function)
    bool host_stale = false;            // ctx->tensors older than d_params
    ConvLayer fnet_dg[14], srnet_dg[24];   // data-gradient convs of the Conv2D layers (planned where needed)
    std::map<std::string, float*> act;  // saved activations and gradient work buffers
    std::map<std::string, size_t> act_img;   // floats per image of each (for forwards over a slice of the batch)
    float* part = nullptr;              // wgrad partial sums, one region per layer (reduced together at the end)
    size_t part_bytes = 0;
    // sr_train_clips_grads: frame-major staging, [n zero frames | frames] so that "previous frame" is a pointer offset
    float* clip_lr = nullptr;           // (2 * max_batch) LR frames
    float* clip_est = nullptr;          // (2 * max_batch) HR frames: n_clips zero frames, then the estimate of every sample
    unsigned char* clip_u8 = nullptr;   // max_batch HR frames as uint8
    double* mse_part = nullptr;
    float* d_loss = nullptr;            // [sr, flow]
    float* h_loss = nullptr;            // the same two values in page-locked host memory (read after a synchronisation)
    // host batches: the two HR-sized inputs are copied on a second stream under FNet / the forward, through page-locked
    // staging filled by helper threads (HostUpload below): [est_tm1 | hr_t] of max_batch samples
    static constexpr int MAX_UP = 8;
    hipStream_t copy_stream[MAX_UP] = {};      // [0] also carries the copies the calling thread issues itself
    hipEvent_t ev_start = nullptr, ev_est[MAX_UP] = {}, ev_hr[MAX_UP] = {};
    unsigned char* h_stage = nullptr;
    int up_threads = 2;                 // SR_H2D_THREADS (<= MAX_UP); 0: plain hipMemcpyAsync from the caller's arrays
};

namespace {

void layer_dims(bool is_f, int i, int& cin, int& cout, int& mode) {
    if (is_f) {
        cin = FNET_SPEC[i][0];
        cout = FNET_SPEC[i][1];
        mode = M_CONV;
    } else {
        srnet_spec(i, cin, cout, mode);
    }
}

int talloc(sr_ctx* c, const char* name, size_t floats) {
    float* p = nullptr;
    HIPCHK(c, hipMalloc((void**)&p, (floats ? floats : 4) * sizeof(float)));
    c->train->act[name] = p;
    c->train->act_img[name] = floats / (size_t)c->max_batch;
    return SR_OK;
}

int upload_ents(sr_ctx* c, const std::vector<PackEnt>& ents, PackEnt** d) {
    if (*d || ents.empty()) return SR_OK;
    HIPCHK(c, hipMalloc((void**)d, ents.size() * sizeof(PackEnt)));
    HIPCHK(c, hipMemcpy(*d, ents.data(), ents.size() * sizeof(PackEnt), hipMemcpyHostToDevice));
    return SR_OK;
}

int upload_params(sr_ctx* c) {
    sr_train_state* T = c->train;
    for (int i = 0; i < 76; ++i)
        HIPCHK(c, hipMemcpy(T->d_params + T->off[i], c->tensors[i].data.data(), c->tensors[i].data.size() * 4, hipMemcpyHostToDevice));
    T->host_stale = false;
    return SR_OK;
}

// first training call: master weights, optimizer state, dgrad layers, activation tape
int train_init(sr_ctx* c) {
    if (c->train) return SR_OK;
    if (c->prec == P_BF16)
        return fail(c, SR_ERR_ARG, "training needs fp32 activations: create the ctx with dtype fp32 or bf16x3");
    sr_train_state* T = new sr_train_state();
    c->train = T;
    for (int i = 0; i < 76; ++i) T->off[i + 1] = T->off[i] + (int64_t)c->tensors[i].data.size();
    T->nparams = T->off[76];
    const size_t nb = (size_t)T->nparams * 4;
    HIPCHK(c, hipMalloc((void**)&T->d_params, nb));
    HIPCHK(c, hipMalloc((void**)&T->d_m, nb));
    HIPCHK(c, hipMalloc((void**)&T->d_v, nb));
    HIPCHK(c, hipMalloc((void**)&T->d_grad, nb));
    HIPCHK(c, hipMemset(T->d_m, 0, nb));
    HIPCHK(c, hipMemset(T->d_v, 0, nb));
    HIPCHK(c, hipMemset(T->d_grad, 0, nb));
    int rc = upload_params(c);
    if (rc) return rc;
    // data-gradient convs (FNet's first layer has no input gradient; the transposed convs use dgrad_s2)
    size_t part_bytes = 0;
    const int h = c->h, w = c->w, B = c->max_batch;
    for (int g = 0; g < 2; ++g) {
        const bool is_f = g == 0;
        const int nl = is_f ? 14 : 24;
        for (int i = 0; i < nl; ++i) {
            int cin, cout, mode;
            layer_dims(is_f, i, cin, cout, mode);
            ConvLayer& L = is_f ? c->fnet[i] : c->srnet[i];
            if ((rc = upload_ents(c, L.ents, &L.d_ents))) return rc;
            if ((rc = upload_ents(c, L.proj_ents, &L.d_proj_ents))) return rc;
            if (L.d_w2 && (rc = upload_ents(c, L.ents2, &L.d_ents2))) return rc;
            if (L.d_w3 && (rc = upload_ents(c, L.ents3, &L.d_ents3))) return rc;
            // spatial size of the layer's input, for the wgrad workspace
            int H = h, W = w;
            if (is_f) {
                const int lvl[14] = {0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 1, 1, 0, 0};
                H = h >> lvl[i];
                W = w >> lvl[i];
            } else if (i == 22) {
                H = 2 * h;
                W = 2 * w;
            } else if (i == 23) {
                H = 4 * h;
                W = 4 * w;
            }
            // every layer keeps its slice partials until the step's single batched reduction
            part_bytes += round_up_sz(wgrad_workspace_bytes(B, H, W, cin, cout, mode == M_CONVT));
            if ((mode == M_CONVT && c->prec != P_BF16X3) || (is_f && i == 0) || cout <= 3) continue;   // <= 3 outputs: dgrad_few reads the master weights
            ConvLayer& D = is_f ? T->fnet_dg[i] : T->srnet_dg[i];
            // the stride-1 data gradient is a conv of the layer's own launch size (H x W is its input = the gradient's output);
            // dgrad_s2_x3 reads the forward packing of a transposed conv and is faster with 64 channels per block: no hint
            plan_layer(D, c->prec, M_CONV, cout, cin, mode == M_CONVT ? 0 : conv_tiles(B, H, W));
            const float* wk = c->tensors[(is_f ? 0 : 28) + 2 * i].data.data();
            // transposed conv (bf16x3): its kernel (kh,kw,out,in) read as HWIO is the stride-2 data-gradient conv's
            // kernel; dgrad_s2_x3 reads it in the forward packing
            if ((rc = upload_layer(c, D, c->prec, wk, nullptr, mode == M_CONVT ? PK_FWD : PK_DGRAD))) return rc;
            if ((rc = upload_ents(c, D.ents, &D.d_ents))) return rc;
        }
    }
    HIPCHK(c, hipMalloc((void**)&T->part, part_bytes));
    T->part_bytes = part_bytes;
    // activation tape and gradient work buffers, fp32, per image sizes in floats
    const size_t L0 = (size_t)h * w, L1 = L0 / 4, L2 = L0 / 16, L3 = L0 / 64, Bz = B;
    struct { const char* n; size_t elems; } plan[] = {
        {"x0", L0 * 16}, {"e0a", L0 * 32}, {"e0b", L0 * 32}, {"p1", L1 * 32}, {"e1a", L1 * 64}, {"e1b", L1 * 64},
        {"p2", L2 * 64}, {"e2a", L2 * 128}, {"e2b", L2 * 128}, {"p3", L3 * 128}, {"e3a", L3 * 256}, {"e3b", L3 * 256},
        {"u2", L2 * 256}, {"d2a", L2 * 128}, {"d2b", L2 * 128}, {"u1", L1 * 128}, {"d1a", L1 * 64}, {"d1b", L1 * 64},
        {"u0", L0 * 64}, {"c13", L0 * 32}, {"s_in", L0 * 64}, {"up1", L0 * 4 * 64}, {"up2", L0 * 16 * 64},
        {"gA", L0 * 16 * 64}, {"gB", L0 * 16 * 64}, {"g_est", L0 * 16 * 4}, {"g_flow", L0 * 16}, {"g_fhr", L0 * 16 * 2},
        {"g_lw", L0 * 3}, {"hr", L0 * 48},
        {"s_out", c->model == SR_TECOGAN ? L0 * 48 : 0}};   // SRNet's clipped output before TecoGAN's tanh epilogue
    for (auto& p : plan)
        if ((rc = talloc(c, p.n, Bz * p.elems))) return rc;
    char nm[16];
    for (int r = 0; r <= 10; ++r) {
        snprintf(nm, sizeof nm, "x%d_", r);
        if ((rc = talloc(c, nm, Bz * L0 * 64))) return rc;
        if (r < 10) {
            snprintf(nm, sizeof nm, "y%d_", r);
            if ((rc = talloc(c, nm, Bz * L0 * 64))) return rc;
        }
    }
    const size_t mse_bytes = std::max(mse_workspace_bytes((long)(Bz * L0 * 16), 4), mse_workspace_bytes((long)(Bz * L0), 3));
    HIPCHK(c, hipMalloc((void**)&T->mse_part, mse_bytes));
    HIPCHK(c, hipMalloc((void**)&T->d_loss, 2 * sizeof(float)));
    HIPCHK(c, hipHostMalloc((void**)&T->h_loss, 2 * sizeof(float), hipHostMallocDefault));
    HIPCHK(c, hipEventCreateWithFlags(&T->ev_start, hipEventDisableTiming));
    if (const char* e = getenv("SR_H2D_THREADS")) T->up_threads = std::max(0, std::min((int)sr_train_state::MAX_UP, atoi(e)));
    for (int k = 0; k < std::max(1, T->up_threads); ++k) {
        HIPCHK(c, hipStreamCreateWithFlags(&T->copy_stream[k], hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&T->ev_est[k], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&T->ev_hr[k], hipEventDisableTiming));
    }
    return SR_OK;
}

struct Tape {
    sr_ctx* c;
    sr_train_state* T;
    int B, P;
    hipStream_t st;
    std::vector<WgradArgs> pending;   // weight gradients whose slice partials wait for reduce_all()
    size_t part_off = 0;
    size_t b0 = 0;                    // first image of the slice of the batch this tape works on (forward only)
    // combined TecoGAN step: the forward also writes the warped previous estimate; the backward starts from a gradient of
    // the output that the caller has already put into "g_est" and adds the discriminator's gradient of that warped estimate
    float* hr_warp_out = nullptr;
    bool ext_g_est = false;
    const float* g27 = nullptr;
    float* a(const char* n) { return T->act[n] + b0 * T->act_img[n]; }

    int conv(const ConvLayer& L, const float* in, int H, int W, float* out, int out_cs, int cout_store, int act,
             const float* res, const char* name) {
        return run_conv(c, P, st, L, B, {in, H, W, out, out_cs, cout_store, 1, res, out_cs, act}, name);
    }
    // data gradient with the derivative of the ReLU / LeakyReLU `act` that produced y folded into the store
    int conv_masked(const ConvLayer& L, const float* in, int H, int W, float* out, int out_cs, int cout_store, int act,
                    const float* y, const char* name) {
        ConvIO io{in, H, W, out, out_cs, cout_store, 1, y, out_cs, act};
        io.res_mask = 1;
        return run_conv(c, P, st, L, B, io, name);
    }
    // kernel + bias gradient of tensor pair `ti` (index of the kernel in file order)
    int wgrad(int ti, int mode, int cin, int cout, const float* x, int xcs, const float* dy, int ycs, int H, int W) {
        WgradArgs wa;
        wa.x = x;
        wa.dy = dy;
        wa.dw = T->d_grad + T->off[ti];
        wa.db = T->d_grad + T->off[ti + 1];
        wa.B = B;
        wa.H = H;
        wa.W = W;
        wa.cin = cin;
        wa.cout = cout;
        wa.xcs = xcs;
        wa.ycs = ycs;
        wa.convt = mode == M_CONVT;
        wa.x3 = P == P_BF16X3;
        wa.defer_reduce = 1;
        const size_t ws = round_up_sz(wgrad_workspace_bytes(B, H, W, cin, cout, wa.convt));
        if (part_off + ws > T->part_bytes) return fail(c, SR_ERR_STATE, "wgrad workspace exhausted");
        wa.part = T->part + part_off / 4;
        part_off += ws;
        pending.push_back(wa);
        const double px = (double)B * H * W;
        char nm[48];
        snprintf(nm, sizeof nm, "train.wgrad%s %d>%d @%dx%d", mode == M_CONVT ? "T" : "", cin, cout, H, W);
        ProfScope ps(c, nm, px * 18.0 * cin * cout, px * 4.0 * (cin + (mode == M_CONVT ? 4 : 1) * cout));
        LAUNCHCHK(c, "wgrad", launch_wgrad(wa, st));
        return SR_OK;
    }
    int reduce_all() {
        ProfScope ps(c, "train.wgrad_reduce", 0, (double)part_off);
        LAUNCHCHK(c, "wgrad_reduce", launch_wgrad_reduce_batch(pending.data(), (int)pending.size(), st));
        pending.clear();
        part_off = 0;
        return SR_OK;
    }
    int mask(float* g, const float* y, int act, long npix, int C, int gcs, int ycs) {
        ProfScope ps(c, "train.act_bwd", 0, (double)npix * C * 12.0);
        LAUNCHCHK(c, "act_bwd", launch_act_bwd(g, y, act, npix, C, gcs, ycs, st));
        return SR_OK;
    }
};

#define RC(call)                  \
    do {                          \
        int rc_ = (call);         \
        if (rc_) return rc_;      \
    } while (0)

// forward of reference FRVSR.py:58-63 with every layer output kept (fp32 NHWC, no planar trunk, no fused tail)
// stage 1: FNet (needs only the two LR frames); stage 2: warps + SRNet (needs the previous estimate)
int train_forward(Tape& t, int stage, const float* d_lr_t, const float* d_lr_tm1, const float* d_est_prev, float* d_est,
                  float* d_lr_warp, float* d_flow) {
    sr_ctx* c = t.c;
    const int B = t.B, H = c->h, W = c->w;
    hipStream_t st = t.st;
    const ConvLayer* F = c->fnet;
    if (stage == 1) {
        LAUNCHCHK(c, "fnet.input", launch_fnet_input(d_lr_t, d_lr_tm1, t.a("x0"), 0, B, H, W, st));
        // encoder
        RC(t.conv(F[0], t.a("x0"), H, W, t.a("e0a"), 32, 32, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[1], t.a("e0a"), H, W, t.a("e0b"), 32, 32, A_LRELU, nullptr, "train.fwd"));
        LAUNCHCHK(c, "pool", launch_maxpool2(t.a("e0b"), t.a("p1"), 0, B, H, W, 32, st));
        RC(t.conv(F[2], t.a("p1"), H / 2, W / 2, t.a("e1a"), 64, 64, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[3], t.a("e1a"), H / 2, W / 2, t.a("e1b"), 64, 64, A_LRELU, nullptr, "train.fwd"));
        LAUNCHCHK(c, "pool", launch_maxpool2(t.a("e1b"), t.a("p2"), 0, B, H / 2, W / 2, 64, st));
        RC(t.conv(F[4], t.a("p2"), H / 4, W / 4, t.a("e2a"), 128, 128, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[5], t.a("e2a"), H / 4, W / 4, t.a("e2b"), 128, 128, A_LRELU, nullptr, "train.fwd"));
        LAUNCHCHK(c, "pool", launch_maxpool2(t.a("e2b"), t.a("p3"), 0, B, H / 4, W / 4, 128, st));
        RC(t.conv(F[6], t.a("p3"), H / 8, W / 8, t.a("e3a"), 256, 256, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[7], t.a("e3a"), H / 8, W / 8, t.a("e3b"), 256, 256, A_LRELU, nullptr, "train.fwd"));
        // decoder
        LAUNCHCHK(c, "resize", launch_resize2x(t.a("e3b"), t.a("u2"), 0, B, H / 8, W / 8, 256, st));
        RC(t.conv(F[8], t.a("u2"), H / 4, W / 4, t.a("d2a"), 128, 128, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[9], t.a("d2a"), H / 4, W / 4, t.a("d2b"), 128, 128, A_LRELU, nullptr, "train.fwd"));
        LAUNCHCHK(c, "resize", launch_resize2x(t.a("d2b"), t.a("u1"), 0, B, H / 4, W / 4, 128, st));
        RC(t.conv(F[10], t.a("u1"), H / 2, W / 2, t.a("d1a"), 64, 64, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[11], t.a("d1a"), H / 2, W / 2, t.a("d1b"), 64, 64, A_LRELU, nullptr, "train.fwd"));
        LAUNCHCHK(c, "resize", launch_resize2x(t.a("d1b"), t.a("u0"), 0, B, H / 2, W / 2, 64, st));
        RC(t.conv(F[12], t.a("u0"), H, W, t.a("c13"), 32, 32, A_LRELU, nullptr, "train.fwd"));
        RC(t.conv(F[13], t.a("c13"), H, W, d_flow, 2, 2, A_TANH, nullptr, "train.fwd"));
        return SR_OK;
    }
    // warps
    LAUNCHCHK(c, "lr_warp", launch_warp3(d_lr_tm1, d_flow, d_lr_warp, B, H, W, st));
    LAUNCHCHK(c, "pre_srnet", launch_pre_srnet(d_flow, d_est_prev, d_lr_t, t.a("s_in"), t.hr_warp_out, 0, B, H, W, st));
    // SRNet
    const ConvLayer* S = c->srnet;
    char xn[16], yn[16], xn1[16];
    RC(t.conv(S[0], t.a("s_in"), H, W, t.a("x0_"), 64, 64, A_RELU, nullptr, "train.fwd"));
    for (int r = 0; r < 10; ++r) {
        snprintf(xn, sizeof xn, "x%d_", r);
        snprintf(yn, sizeof yn, "y%d_", r);
        snprintf(xn1, sizeof xn1, "x%d_", r + 1);
        RC(t.conv(S[1 + 2 * r], t.a(xn), H, W, t.a(yn), 64, 64, A_RELU, nullptr, "train.fwd"));
        RC(t.conv(S[2 + 2 * r], t.a(yn), H, W, t.a(xn1), 64, 64, A_NONE, t.a(xn), "train.fwd"));
    }
    RC(t.conv(S[21], t.a("x10_"), H, W, t.a("up1"), 64, 64, A_RELU, nullptr, "train.fwd"));
    RC(t.conv(S[22], t.a("up1"), 2 * H, 2 * W, t.a("up2"), 64, 64, A_RELU, nullptr, "train.fwd"));
    if (c->model == SR_TECOGAN) {
        // generator output = tanh(SRNet + bilinear x4 (LR_t)) (reference TecoGAN.py:74-76); the clipped SRNet
        // output is kept for the clip's mask
        RC(t.conv(S[23], t.a("up2"), 4 * H, 4 * W, t.a("s_out"), 3, 3, A_CLIP01, nullptr, "train.fwd"));
        HIPCHK(c, hipMemcpyAsync(d_est, t.a("s_out"), (size_t)B * H * W * 48 * 4, hipMemcpyDeviceToDevice, st));
        LAUNCHCHK(c, "teco_epilogue", launch_teco_epilogue(d_est, d_lr_t, B, H, W, st));
        return SR_OK;
    }
    RC(t.conv(S[23], t.a("up2"), 4 * H, 4 * W, d_est, 3, 3, A_CLIP01, nullptr, "train.fwd"));
    return SR_OK;
}

int train_backward(Tape& t, const float* d_lr_t, const float* d_lr_tm1, const float* d_est_prev, const float* d_hr,
                   const float* d_est, const float* d_lr_warp, const float* d_flow) {
    sr_ctx* c = t.c;
    sr_train_state* T = t.T;
    const int B = t.B, H = c->h, W = c->w;
    hipStream_t st = t.st;
    const long L0 = (long)B * H * W;
    float *cur = t.a("gA"), *oth = t.a("gB");
    auto swap = [&]() { std::swap(cur, oth); };
    const ConvLayer* SD = T->srnet_dg;
    const int so = 28;   // index of SRNet's first tensor
    // ---- losses (Keras: mean over all elements) and their gradients
    if (!t.ext_g_est)
        LAUNCHCHK(c, "mse", launch_mse_grad(d_est, d_hr, t.a("g_est"), T->mse_part, T->d_loss, 16 * L0, 3, 4,
                                               c->model == SR_TECOGAN ? t.a("s_out") : d_est, c->model == SR_TECOGAN, st));
    // ---- SRNet, reverse order (reference layers.py:66-76)
    RC(t.wgrad(so + 46, M_CONV, 64, 3, t.a("up2"), 64, t.a("g_est"), 4, 4 * H, 4 * W));
    {   // data gradient of the 64->3 conv with up2's ReLU mask in the store: one pass over the step's largest tensor
        ProfScope ps(c, "train.dgrad_few", 16.0 * L0 * 54 * 64, 16.0 * L0 * 4 * (64 + 64 + 3));
        LAUNCHCHK(c, "dgrad_few", launch_dgrad_few(t.a("g_est"), T->d_params + T->off[so + 46], cur, t.a("up2"), A_RELU, B, 4 * H, 4 * W,
                                                   64, 3, 4, 64, st));
    }
    RC(t.wgrad(so + 44, M_CONVT, 64, 64, t.a("up1"), 64, cur, 64, 2 * H, 2 * W));
    {
        ProfScope ps(c, "train.dgrad_s2", 4.0 * L0 * 18 * 64 * 64, 4.0 * L0 * 4 * (4 * 64 + 64));
        if (t.P == P_BF16X3)
            LAUNCHCHK(c, "dgrad_s2", launch_dgrad_s2_x3(cur, SD[22].d_w, SD[22].nt, SD[22].wstride_cb, oth, t.a("up1"), B, 2 * H, 2 * W, 64, 64, 64, 64, st));
        else
            LAUNCHCHK(c, "dgrad_s2", launch_dgrad_s2(cur, T->d_params + T->off[so + 44], oth, t.a("up1"), B, 2 * H, 2 * W, 64, 64, 64, 64, st));
    }
    swap();   // ReLU mask of up1 applied in the kernel's store
    RC(t.wgrad(so + 42, M_CONVT, 64, 64, t.a("x10_"), 64, cur, 64, H, W));
    {
        ProfScope ps(c, "train.dgrad_s2", 1.0 * L0 * 18 * 64 * 64, 1.0 * L0 * 4 * (4 * 64 + 64));
        if (t.P == P_BF16X3)
            LAUNCHCHK(c, "dgrad_s2", launch_dgrad_s2_x3(cur, SD[21].d_w, SD[21].nt, SD[21].wstride_cb, oth, nullptr, B, H, W, 64, 64, 64, 64, st));
        else
            LAUNCHCHK(c, "dgrad_s2", launch_dgrad_s2(cur, T->d_params + T->off[so + 42], oth, nullptr, B, H, W, 64, 64, 64, 64, st));
    }
    swap();   // cur = g_x
    char xn[16], yn[16];
    for (int r = 9; r >= 0; --r) {
        snprintf(xn, sizeof xn, "x%d_", r);
        snprintf(yn, sizeof yn, "y%d_", r);
        const int l2 = 2 + 2 * r, l1 = 1 + 2 * r;
        RC(t.wgrad(so + 2 * l2, M_CONV, 64, 64, t.a(yn), 64, cur, 64, H, W));
        RC(t.conv_masked(SD[l2], cur, H, W, oth, 64, 64, A_RELU, t.a(yn), "train.dgrad"));
        RC(t.wgrad(so + 2 * l1, M_CONV, 64, 64, t.a(xn), 64, oth, 64, H, W));
        RC(t.conv(SD[l1], oth, H, W, cur, 64, 64, A_NONE, cur, "train.dgrad"));   // g_x += dgrad, in place like the forward
    }
    RC(t.mask(cur, t.a("x0_"), A_RELU, L0, 64, 64, 64));
    RC(t.wgrad(so + 0, M_CONV, 51, 64, t.a("s_in"), 64, cur, 64, H, W));
    RC(t.conv(SD[0], cur, H, W, oth, 64, 51, A_NONE, nullptr, "train.dgrad"));
    swap();   // cur = gradient of SRNet's input [.., 64], channels 3..50 = space_to_depth(warped estimate)
    // ---- warps: gradient w.r.t. the flow only (the previous estimate and LR frame are constants)
    {
        ProfScope ps(c, "train.warp_bwd", 0, (double)L0 * 16 * (12 + 12 + 8) + (double)L0 * 256);
        if (t.g27) LAUNCHCHK(c, "add_hat_s2d", launch_add_hat_s2d(t.g27, cur, B, H, W, 64, st));
        LAUNCHCHK(c, "hr_warp_bwd", launch_hr_warp_bwd(d_est_prev, d_flow, cur, t.a("g_fhr"), B, H, W, 64, st));
        HIPCHK(c, hipMemsetAsync(t.a("g_flow"), 0, (size_t)L0 * 16 * 4, st));
        LAUNCHCHK(c, "resize_bwd", launch_resize_bwd(t.a("g_fhr"), t.a("g_flow"), B, H, W, 2, 4, 16, 0, nullptr, A_NONE, st));
        LAUNCHCHK(c, "mse", launch_mse_grad(d_lr_warp, d_lr_t, t.a("g_lw"), T->mse_part, T->d_loss + 1, L0, 3, 3, nullptr, 0, st));
        LAUNCHCHK(c, "warp_bwd_flow", launch_warp_bwd_flow(d_lr_tm1, d_flow, t.a("g_lw"), t.a("g_flow"), B, H, W, 3, 16, 1, st));
    }
    RC(t.mask(t.a("g_flow"), d_flow, A_TANH, L0, 2, 16, 2));
    // ---- FNet, reverse order (reference layers.py:27-45)
    const ConvLayer* FD = T->fnet_dg;
    // every gradient below leaves its kernel already multiplied by the LeakyReLU derivative of the tensor it
    // lands on (x is a LeakyReLU output in all three helpers), so no separate mask pass runs in FNet
    auto pool_bwd = [&](const char* x, int h, int w, int C) -> int {   // gradient of x's 2x2 pooled output in cur -> oth
        ProfScope ps(c, "train.pool_bwd", 0, (double)B * h * w * C * 4 * 2.25);
        LAUNCHCHK(c, "maxpool2_bwd", launch_maxpool2_bwd(t.a(x), cur, oth, B, h, w, C, A_LRELU, st));
        return SR_OK;
    };
    auto resize_bwd = [&](const char* x, int h, int w, int C) -> int {   // cur [B,2h,2w,C] -> oth [B,h,w,C] = d/dx
        ProfScope ps(c, "train.resize_bwd", 0, (double)B * h * w * C * 4 * 6.0);
        LAUNCHCHK(c, "resize_bwd", launch_resize_bwd(cur, oth, B, h, w, C, 2, C, 0, t.a(x), A_LRELU, st));
        return SR_OK;
    };
    // one conv layer: dy = dL/d(out) with the activation mask already applied; x_is_act: the layer's input is a
    // LeakyReLU output (not a pooled / resized tensor), so its derivative goes into the data-gradient's store
    auto conv_bwd = [&](int i, const char* x, int xcs, const float* dy, int ycs, int h, int w, float* dx,
                        bool x_is_act = false) -> int {
        const int cin = FNET_SPEC[i][0], cout = FNET_SPEC[i][1];
        RC(t.wgrad(2 * i, M_CONV, cin, cout, t.a(x), xcs, dy, ycs, h, w));
        if (!dx) return SR_OK;
        if (x_is_act) return t.conv_masked(FD[i], dy, h, w, dx, cin, cin, A_LRELU, t.a(x), "train.dgrad");
        return t.conv(FD[i], dy, h, w, dx, cin, cin, A_NONE, nullptr, "train.dgrad");
    };
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8;
    RC(conv_bwd(13, "c13", 32, t.a("g_flow"), 16, H, W, nullptr));
    {   // flow head 32->2: data gradient with c13's LeakyReLU derivative in the store
        ProfScope ps(c, "train.dgrad_few", 1.0 * L0 * 36 * 32, 1.0 * L0 * 4 * (32 + 32 + 2));
        LAUNCHCHK(c, "dgrad_few", launch_dgrad_few(t.a("g_flow"), T->d_params + T->off[26], cur, t.a("c13"), A_LRELU, B, H, W, 32, 2, 16, 32, st));
    }
    RC(conv_bwd(12, "u0", 64, cur, 32, H, W, oth));
    swap();
    RC(resize_bwd("d1b", H1, W1, 64));
    swap();
    RC(conv_bwd(11, "d1a", 64, cur, 64, H1, W1, oth, true));
    swap();
    RC(conv_bwd(10, "u1", 128, cur, 64, H1, W1, oth));
    swap();
    RC(resize_bwd("d2b", H2, W2, 128));
    swap();
    RC(conv_bwd(9, "d2a", 128, cur, 128, H2, W2, oth, true));
    swap();
    RC(conv_bwd(8, "u2", 256, cur, 128, H2, W2, oth));
    swap();
    RC(resize_bwd("e3b", H3, W3, 256));
    swap();
    RC(conv_bwd(7, "e3a", 256, cur, 256, H3, W3, oth, true));
    swap();
    RC(conv_bwd(6, "p3", 128, cur, 256, H3, W3, oth));
    swap();
    RC(pool_bwd("e2b", H2, W2, 128));
    swap();
    RC(conv_bwd(5, "e2a", 128, cur, 128, H2, W2, oth, true));
    swap();
    RC(conv_bwd(4, "p2", 64, cur, 128, H2, W2, oth));
    swap();
    RC(pool_bwd("e1b", H1, W1, 64));
    swap();
    RC(conv_bwd(3, "e1a", 64, cur, 64, H1, W1, oth, true));
    swap();
    RC(conv_bwd(2, "p1", 32, cur, 64, H1, W1, oth));
    swap();
    RC(pool_bwd("e0b", H, W, 32));
    swap();
    RC(conv_bwd(1, "e0a", 32, cur, 32, H, W, oth, true));
    swap();
    RC(conv_bwd(0, "x0", 16, cur, 32, H, W, nullptr));
    return t.reduce_all();
}

// The two HR-sized HOST inputs of a batch (previous estimate, then the HR target: 31 MB each at 40 x 256 x 256) on their way to
// the device.  hipMemcpyAsync from PAGEABLE memory blocks the calling thread for the whole transfer -- one thread's memcpy into
// the runtime's staging, 1.03 - 1.1 ms per tensor -- so, issued behind FNet's launches, the first copy left the GPU idle for
// 1.1 ms of every step (FNet of a 40 x 64 x 64 batch takes 0.25 ms) and the second held the backward's launches back (another
// 0.3 ms in front of the loss).  Measured alternatives: helper threads issuing the pageable copies (they serialise inside the
// runtime: no gain with 1, slower with 2 / 4); one thread's memcpy into own page-locked staging (as slow as the runtime's);
// many 1 MiB DMAs on one stream (consecutive copies of a stream start ~25 us apart whatever their size: 20 GB/s).
// What is built: `nw` helper threads, started before anything else, thread k on copy stream k.  Each memcpys its slice of the
// estimate, then of the target, into page-locked staging in 4 MiB pieces and hands every piece to its stream as an
// asynchronous DMA; the calling thread launches kernels meanwhile and calls done() where the ctx stream needs a tensor (that
// waits until the helpers have ISSUED the stage -- an event must be recorded before a stream can wait for it --, records the
// events and makes the ctx stream wait for them).  Two threads: the estimate is on the device 0.55 - 0.7 ms after the call
// began (PCIe: 0.56 ms) and the step takes 8.7 instead of 9.2 ms; 3 or 4 threads measured slower again (8.9 - 9.1).
// Helpers only report errors (no ctx access); the counters always reach `nw`, so done() cannot wait forever.
struct HostUpload {
    struct Job {
        char* dst;
        const char* src;
        char* stage;
    };
    int nw = 0;
    std::thread th[8];
    std::mutex m;
    std::condition_variable cv;
    int finished[2] = {0, 0};      // helpers done with job 0 (estimate) / job 1 (target)
    hipError_t err = hipSuccess;

    void start(int device, const hipStream_t* streams, hipEvent_t after, int n_threads, Job j0, Job j1, size_t bytes) {
        nw = n_threads;
        for (int k = 0; k < nw; ++k) {
            const size_t lo = (bytes / 4096 * k / nw) * 4096, hi = k + 1 == nw ? bytes : (bytes / 4096 * (k + 1) / nw) * 4096;
            hipStream_t cs = streams[k];
            th[k] = std::thread([this, device, cs, after, j0, j1, lo, hi] {
                hipError_t e = hipSetDevice(device);
                if (e == hipSuccess) e = hipStreamWaitEvent(cs, after, 0);
                const Job jobs[2] = {j0, j1};
                for (int j = 0; j < 2; ++j) {
                    // few, large DMAs: consecutive copies of one stream start ~25 us apart whatever their size
                    const size_t piece = (size_t)4 << 20;
                    for (size_t o = lo; o < hi && e == hipSuccess && jobs[j].src; o += piece) {
                        const size_t len = std::min(piece, hi - o);
                        memcpy(jobs[j].stage + o, jobs[j].src + o, len);
                        e = hipMemcpyAsync(jobs[j].dst + o, jobs[j].stage + o, len, hipMemcpyHostToDevice, cs);
                    }
                    {
                        std::lock_guard<std::mutex> g(m);
                        if (e != hipSuccess && err == hipSuccess) err = e;
                        ++finished[j];
                    }
                    cv.notify_all();
                }
            });
        }
    }
    // every piece of job j is on its copy stream: record the events behind them and make `st` wait for these
    hipError_t done(int j, const hipStream_t* streams, hipEvent_t* evs, hipStream_t st) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return finished[j] == nw; });
        hipError_t e = err;
        l.unlock();
        for (int k = 0; k < nw && e == hipSuccess; ++k) {
            e = hipEventRecord(evs[k], streams[k]);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, evs[k], 0);
        }
        return e;
    }
    ~HostUpload() {
        for (int k = 0; k < nw; ++k)
            if (th[k].joinable()) th[k].join();
    }
};

// sr_train_on_batch / sr_train_on_clips enqueue the optimizer step behind the backward pass BEFORE they wait for the losses
// (waiting in between left the GPU idle for ~50 us of every step: the host's wake-up plus the Adam launch)
thread_local bool g_defer_loss_sync = false;

int finish_losses(sr_ctx* c, float* losses) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const float* l = c->train->h_loss;
    if (losses) {   // Keras train_on_batch order: [total, output 1, output 2] (reference FRVSR.py:161)
        losses[0] = l[0] + l[1];
        losses[1] = l[0];
        losses[2] = l[1];
    }
    return SR_OK;
}

// n_clips > 0 (sr_train_clips_grads): est_tm1 is the frame-major estimate buffer, n_clips zero frames followed by room
// for the estimates of all B samples, which the forward fills frame group by frame group
int grads_common(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, const float* hr_t,
                 float* d_grads_out, float* losses, bool dev, int n_clips = 0) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (c->pad) return fail(c, SR_ERR_STATE, "training needs a ctx whose tile size is a multiple of 8 (sr_create, not sr_create_padded)");
    if (!lr_t || !lr_tm1 || !est_tm1 || !hr_t) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    if ((rc = train_init(c))) {
        train_free(c);   // a half-built state must not survive
        return rc;
    }
    sr_train_state* T = c->train;
    c->clip_active = false;   // borrows the state buffers
    const size_t nlr = (size_t)B * c->h * c->w * 3 * 4, nhr = nlr * 16;
    hipStream_t st = c->stream;
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // The two HR-sized inputs (31 MB each at 40 x 256 x 256) go over a second stream: FNet runs under the copy of the
    // previous estimate, the rest of the forward under the copy of the HR target (host arrays: HostUpload above).
    hipStream_t cs = T->copy_stream[0];
    HIPCHK(c, hipEventRecord(T->ev_start, st));                 // earlier work on the ctx stream may still read the buffers
    HostUpload up;                                              // joins its threads on every way out of this function
    const bool staged = !dev && n_clips == 0 && T->up_threads > 0 && nhr >= ((size_t)1 << 20);
    if (staged) {
        const size_t cap_hr = (size_t)c->max_batch * c->h * c->w * 48 * 4;
        if (!T->h_stage) HIPCHK(c, hipHostMalloc((void**)&T->h_stage, 2 * cap_hr + cap_hr / 8, hipHostMallocDefault));   // + the two LR tensors
        // the staging is free again when this call returns: it ends with a synchronisation behind every copy
        up.start(c->device, T->copy_stream, T->ev_start, T->up_threads, {(char*)c->d_est[0], (const char*)est_tm1, (char*)T->h_stage},
                 {(char*)T->act["hr"], hr_t != T->act["hr"] ? (const char*)hr_t : nullptr, (char*)T->h_stage + cap_hr}, nhr);
    } else {
        HIPCHK(c, hipStreamWaitEvent(cs, T->ev_start, 0));
    }
    if (staged) {
        // the two LR tensors the same way from this thread (as pageable copies they held FNet's first launch back by 0.2 ms)
        unsigned char* h_lr = T->h_stage + 2 * (size_t)c->max_batch * c->h * c->w * 48 * 4;
        memcpy(h_lr, lr_t, nlr);
        HIPCHK(c, hipMemcpyAsync(c->d_lr_t, h_lr, nlr, hipMemcpyHostToDevice, st));
        memcpy(h_lr + nlr, lr_tm1, nlr);
        HIPCHK(c, hipMemcpyAsync(c->d_lr_tm1, h_lr + nlr, nlr, hipMemcpyHostToDevice, st));
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_lr_t, lr_t, nlr, kind, st));
        HIPCHK(c, hipMemcpyAsync(c->d_lr_tm1, lr_tm1, nlr, kind, st));
    }
    Tape t{c, T, B, c->prec, st};
    if ((rc = train_forward(t, 1, c->d_lr_t, c->d_lr_tm1, c->d_est[0], c->d_est[1], c->d_lr_warp, c->d_flow))) return rc;
    const float* est_prev_all = c->d_est[0];
    float* est_all = c->d_est[1];
    if (n_clips > 0) {
        // The no-gradient recurrence (reference FRVSR.py:124-151), est_f = G(lr_f, lr_f-1, est_f-1), IS the training
        // forward of frame f's samples, so the taped forward runs frame group by frame group (a contiguous slice of
        // every tape tensor in the frame-major layout) and each group's estimate is the next group's previous
        // estimate: no separate recurrence.  FNet of all groups is the batched pass that just ran.
        float* E = const_cast<float*>(est_tm1);          // [n_clips zero frames | estimates of all frames]
        const size_t fl = (size_t)c->h * c->w, fh = fl * 48;
        est_prev_all = E;
        est_all = E + (size_t)n_clips * fh;
        for (int f = 0; f < B / n_clips; ++f) {
            const size_t o = (size_t)f * n_clips;
            Tape g{c, T, n_clips, c->prec, st};
            g.b0 = o;
            if ((rc = train_forward(g, 2, c->d_lr_t + o * fl * 3, c->d_lr_tm1 + o * fl * 3, E + o * fh, E + (o + n_clips) * fh,
                                    c->d_lr_warp + o * fl * 3, c->d_flow + o * fl * 2)))
                return rc;
        }
    } else {
        if (staged) {
            HIPCHK(c, up.done(0, T->copy_stream, T->ev_est, st));
        } else {
            HIPCHK(c, hipMemcpyAsync(c->d_est[0], est_tm1, nhr, kind, cs));
            HIPCHK(c, hipEventRecord(T->ev_est[0], cs));
            HIPCHK(c, hipStreamWaitEvent(st, T->ev_est[0], 0));
        }
            if ((rc = train_forward(t, 2, c->d_lr_t, c->d_lr_tm1, c->d_est[0], c->d_est[1], c->d_lr_warp, c->d_flow))) return rc;
    }
    if (staged) {
        HIPCHK(c, up.done(1, T->copy_stream, T->ev_hr, st));
    } else {
        if (hr_t != T->act["hr"]) HIPCHK(c, hipMemcpyAsync(T->act["hr"], hr_t, nhr, kind, cs));
        HIPCHK(c, hipEventRecord(T->ev_hr[0], cs));
        HIPCHK(c, hipStreamWaitEvent(st, T->ev_hr[0], 0));
    }
    if ((rc = train_backward(t, c->d_lr_t, c->d_lr_tm1, est_prev_all, T->act["hr"], est_all, c->d_lr_warp, c->d_flow))) return rc;
    if (d_grads_out) HIPCHK(c, hipMemcpyAsync(d_grads_out, T->d_grad, (size_t)T->nparams * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(c, hipMemcpyAsync(T->h_loss, T->d_loss, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (g_defer_loss_sync) return SR_OK;      // sr_train_on_batch / sr_train_on_clips: the Adam step is enqueued first
    return finish_losses(c, losses);
}

// packed MFMA buffers of every conv (forward, data gradient, fused tap projection) from the master weights
int repack_all(sr_ctx* c) {
    sr_train_state* T = c->train;
    hipStream_t st = c->stream;
    RepackBatch rb;
    rb.njobs = 0;
    auto flush = [&]() -> int {
        ProfScope ps(c, "train.repack", 0, 0);
        LAUNCHCHK(c, "repack", launch_repack_batch(T->d_params, rb, c->prec, st));
        rb.njobs = 0;
        return SR_OK;
    };
    auto add = [&](const void* ents, int n, void* out, size_t src) -> int {
        if (n <= 0 || !out) return SR_OK;
        if (rb.njobs == RepackBatch::MAXJ) RC(flush());
        rb.j[rb.njobs++] = {ents, out, n, (int)src, 0, 0};
        return SR_OK;
    };
    for (int g = 0; g < 2; ++g) {
        const bool is_f = g == 0;
        const int nl = is_f ? 14 : 24, base = is_f ? 0 : 28;
        for (int i = 0; i < nl; ++i) {
            ConvLayer& L = is_f ? c->fnet[i] : c->srnet[i];
            const size_t wk = T->off[base + 2 * i];
            RC(add(L.d_ents, (int)L.ents.size(), L.d_w, wk));
            if (L.d_w2) RC(add(L.d_ents2, (int)L.ents2.size(), L.d_w2, wk));   // split-planar trunk packing
            if (L.d_w3) RC(add(L.d_ents3, (int)L.ents3.size(), L.d_w3, wk));   // conv_tail.hip's packing of conv2d_transpose_2
            RC(add(nullptr, L.cout, L.d_bias, T->off[base + 2 * i + 1]));
            ConvLayer& D = is_f ? T->fnet_dg[i] : T->srnet_dg[i];
            if (D.d_w) RC(add(D.d_ents, (int)D.ents.size(), D.d_w, wk));
        }
    }
    ConvLayer& U = c->srnet[22];
    if (U.d_proj) RC(add(U.d_proj_ents, (int)U.proj_ents.size(), U.d_proj, T->off[28 + 46]));
    return flush();
}

}  // namespace

namespace srh {

void train_free(sr_ctx* c) {
    sr_train_state* T = c->train;
    if (!T) return;
    float* fl[] = {T->d_params, T->d_m, T->d_v, T->d_grad, T->part, T->d_loss, T->d_m2, T->d_v2};
    for (auto p : fl)
        if (p) (void)hipFree(p);
    if (T->mse_part) (void)hipFree(T->mse_part);
    for (void* p : {(void*)T->clip_lr, (void*)T->clip_est, (void*)T->clip_u8})
        if (p) (void)hipFree(p);
    for (hipStream_t cs : T->copy_stream)
        if (cs) {
            (void)hipStreamSynchronize(cs);
            (void)hipStreamDestroy(cs);
        }
    if (T->h_stage) (void)hipHostFree(T->h_stage);
    if (T->h_loss) (void)hipHostFree(T->h_loss);
    if (T->ev_start) (void)hipEventDestroy(T->ev_start);
    for (int k = 0; k < sr_train_state::MAX_UP; ++k)
        for (hipEvent_t e : {T->ev_est[k], T->ev_hr[k]})
            if (e) (void)hipEventDestroy(e);
    for (auto& kv : T->act) (void)hipFree(kv.second);
    for (auto& L : T->fnet_dg) free_layer(L);
    for (auto& L : T->srnet_dg) free_layer(L);
    delete T;
    c->train = nullptr;
}

int train_sync_host(sr_ctx* c) {
    sr_train_state* T = c->train;
    if (!T || !T->host_stale) return SR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 76; ++i)
        HIPCHK(c, hipMemcpy(c->tensors[i].data.data(), T->d_params + T->off[i], c->tensors[i].data.size() * 4, hipMemcpyDeviceToHost));
    T->host_stale = false;
    return SR_OK;
}

// a new load_weights (sr_commit_weights) replaces the master weights; Keras keeps the optimizer state
int train_reload(sr_ctx* c) {
    if (!c->train) return SR_OK;
    int rc = upload_params(c);
    if (rc) return rc;
    return repack_all(c);   // the dgrad and projection buffers follow the new weights
}

}  // namespace srh

extern "C" {

int sr_train_param_count(const sr_ctx* c, int64_t* n) {
    if (!c || !n) return fail(nullptr, SR_ERR_ARG, "NULL argument");
    if (!c->committed) return fail(nullptr, SR_ERR_WEIGHTS, "weights not committed");
    int64_t s = 0;
    for (const auto& t : c->tensors) s += (int64_t)t.data.size();
    *n = s;
    return SR_OK;
}

int sr_train_grads(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, const float* hr_t,
                   float* d_grads, float* losses) {
    return grads_common(c, B, lr_t, lr_tm1, est_tm1, hr_t, d_grads, losses, false);
}

int sr_train_grads_dev(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, const float* hr_t,
                       float* d_grads, float* losses) {
    return grads_common(c, B, lr_t, lr_tm1, est_tm1, hr_t, d_grads, losses, true);
}

int sr_train_read_grads(sr_ctx* c, float* host_out) {
    if (!c || !host_out) return fail(c, SR_ERR_ARG, "NULL argument");
    if (!c->train) return fail(c, SR_ERR_STATE, "no gradients: call sr_train_grads first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host_out, c->train->d_grad, (size_t)c->train->nparams * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_train_allreduce(sr_ctx* c, void* comm, float* d_grads) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!comm) return fail(c, SR_ERR_ARG, "comm is NULL");
    if (!c->train) return fail(c, SR_ERR_STATE, "no gradients: call sr_train_grads first");
    HIPCHK(c, hipSetDevice(c->device));
    return rccl_allreduce_f32(c, comm, d_grads ? d_grads : c->train->d_grad, (size_t)c->train->nparams, c->stream);
}

int sr_train_apply(sr_ctx* c, const float* d_grads, float grad_scale, float lr, float beta1, float beta2, float eps) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->train) return fail(c, SR_ERR_STATE, "no gradients: call sr_train_grads first");
    HIPCHK(c, hipSetDevice(c->device));
    sr_train_state* T = c->train;
    T->t += 1;
    // Keras 2.2.4 Adam.get_updates: lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)T->t)) / (1.0 - std::pow((double)beta1, (double)T->t));
    LAUNCHCHK(c, "adam", launch_adam(T->d_params, T->d_m, T->d_v, d_grads ? d_grads : T->d_grad, T->nparams, (float)lr_t, beta1,
                                     beta2, eps, grad_scale, c->stream));
    T->host_stale = true;
    c->mx6_valid = false;   // SR_F16MX6: the host-packed conv_trunk6.hip weights are stale; steps run on the bf16x3 kernels
    return repack_all(c);
}

int sr_train_on_batch(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, const float* hr_t,
                      float lr, float* losses) {
    g_defer_loss_sync = true;
    int rc = grads_common(c, B, lr_t, lr_tm1, est_tm1, hr_t, nullptr, losses, false);
    g_defer_loss_sync = false;
    if (rc) return rc;
    if ((rc = sr_train_apply(c, nullptr, 1.f, lr, 0.9f, 0.999f, 1e-7f))) return rc;
    return finish_losses(c, losses);
}

// Training iteration from HR clips (reference FRVSR.py:103-156 without its per-sample host round trips):
// the HR frames go to the device once (uint8 or float), LR synthesis, the no-gradient recurrence that makes
// the previous estimates, and forward + backward all run on device-resident tensors.  Internally the batch is
// frame-major (sample = frame * n_clips + clip): a step of the recurrence is then one contiguous batch and
// "previous LR frame / previous estimate" are the same buffers n_clips frames earlier, behind n_clips zero
// frames.  The loss and the gradient are sums over the samples, so the order does not change them.
int sr_train_clips_grads(sr_ctx* c, int n_clips, int c_frames, const void* hr, int hr_is_u8, float* d_grads, float* losses) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!hr) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    if (n_clips < 1 || c_frames < 1) return fail(c, SR_ERR_ARG, "n_clips and c_frames must be positive");
    const int B = n_clips * c_frames;
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (c->pad) return fail(c, SR_ERR_STATE, "training needs a ctx whose tile size is a multiple of 8 (sr_create, not sr_create_padded)");
    if ((rc = train_init(c))) {
        train_free(c);
        return rc;
    }
    sr_train_state* T = c->train;
    const size_t fl = (size_t)c->h * c->w * 3, fh = fl * 16;   // floats per LR / HR frame
    if (!T->clip_lr) {
        HIPCHK(c, hipMalloc((void**)&T->clip_lr, 2 * (size_t)c->max_batch * fl * 4));
        HIPCHK(c, hipMalloc((void**)&T->clip_est, 2 * (size_t)c->max_batch * fh * 4));
        HIPCHK(c, hipMalloc((void**)&T->clip_u8, (size_t)c->max_batch * fh));
    }
    c->clip_active = false;   // the recurrence borrows the ctx's work buffers
    hipStream_t st = c->stream;
    float* d_hr = T->act["hr"];
    float* d_lr = T->clip_lr + (size_t)n_clips * fl;
    // clip-major host frames -> frame-major device frames: one strided copy per clip
    const size_t esz = hr_is_u8 ? 1 : 4;
    unsigned char* dst0 = hr_is_u8 ? T->clip_u8 : (unsigned char*)d_hr;
    for (int k = 0; k < n_clips; ++k)
        HIPCHK(c, hipMemcpy2DAsync(dst0 + (size_t)k * fh * esz, (size_t)n_clips * fh * esz,
                                   (const unsigned char*)hr + (size_t)k * c_frames * fh * esz, fh * esz, fh * esz, c_frames,
                                   hipMemcpyHostToDevice, st));
    if (hr_is_u8) LAUNCHCHK(c, "u8_normalize", launch_u8_normalize(T->clip_u8, d_hr, (size_t)B * fh, st));
    HIPCHK(c, hipMemsetAsync(T->clip_lr, 0, (size_t)n_clips * fl * 4, st));
    HIPCHK(c, hipMemsetAsync(T->clip_est, 0, (size_t)n_clips * fh * 4, st));
    LAUNCHCHK(c, "blur_cubic_down4", launch_blur_cubic_down4(d_hr, d_lr, B, 4 * c->h, 4 * c->w, 3, st));
    return grads_common(c, B, d_lr, T->clip_lr, T->clip_est, d_hr, d_grads, losses, true, n_clips);
}

int sr_train_on_clips(sr_ctx* c, int n_clips, int c_frames, const void* hr, int hr_is_u8, float lr, float* losses) {
    g_defer_loss_sync = true;
    int rc = sr_train_clips_grads(c, n_clips, c_frames, hr, hr_is_u8, nullptr, losses);
    g_defer_loss_sync = false;
    if (rc) return rc;
    if ((rc = sr_train_apply(c, nullptr, 1.f, lr, 0.9f, 0.999f, 1e-7f))) return rc;
    return finish_losses(c, losses);
}

int64_t sr_train_iterations(const sr_ctx* c) { return (c && c->train) ? c->train->t : 0; }

}  // extern "C"

// ---------------------------------------------------------------- single backward ops (parity tests)
namespace {
struct DevBufs {
    std::vector<void*> p;
    ~DevBufs() {
        for (auto q : p) (void)hipFree(q);
    }
    float* get(size_t floats) {
        void* q = nullptr;
        if (hipMalloc(&q, (floats ? floats : 4) * 4) != hipSuccess) return nullptr;
        p.push_back(q);
        return (float*)q;
    }
    float* put(const float* host, size_t floats) {
        float* q = get(floats);
        if (q && hipMemcpy(q, host, floats * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return q;
    }
};

int bwd_prologue(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SR_ERR_HIP, "no HIP device available; libsr_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, SR_ERR_ARG, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SR_ERR_HIP, "hipSetDevice failed");
    return SR_OK;
}

int conv_bwd_op(int device, int dtype, int convt, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                const float* dy, float* dx, float* dw, float* db) {
    int rc = bwd_prologue(device);
    if (rc) return rc;
    if (!x || !w || !dy || !dw) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    if (dtype != SR_FP32 && dtype != SR_BF16X3) return fail(nullptr, SR_ERR_ARG, "backward ops need dtype fp32 or bf16x3");
    if (convt && dx && (Cout % 64 || Cin % 32 || (Cin > 32 && Cin % 64)))
        return fail(nullptr, SR_ERR_ARG, "transposed-conv data gradient needs Cout %% 64 == 0 and Cin 32 or a multiple of 64");
    DevBufs d;
    const size_t npix = (size_t)B * H * W, opix = convt ? 4 * npix : npix, nw = (size_t)9 * Cin * Cout;
    float* d_x = d.put(x, npix * Cin);
    float* d_dy = d.put(dy, opix * Cout);
    float* d_w = d.put(w, nw);
    float* d_dw = d.get(nw);
    float* d_db = d.get(Cout);
    float* d_part = d.get(wgrad_workspace_bytes(B, H, W, Cin, Cout, convt) / 4);
    if (!d_x || !d_dy || !d_w || !d_dw || !d_db || !d_part) return fail(nullptr, SR_ERR_HIP, "hipMalloc / copy failed");
    WgradArgs wa = {d_x, d_dy, d_part, d_dw, d_db, B, H, W, Cin, Cout, Cin, Cout, convt, dtype == SR_BF16X3};
    LAUNCHCHK(nullptr, "wgrad", launch_wgrad(wa, nullptr));
    if (dx) {
        float* d_dx = d.get(npix * Cin);
        if (!d_dx) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
        if (convt && dtype == SR_BF16X3) {
            ConvLayer D;
            plan_layer(D, dtype, M_CONV, Cout, Cin);
            struct Guard { ConvLayer& l; ~Guard() { free_layer(l); } } guard{D};
            if ((rc = upload_layer(nullptr, D, dtype, w, nullptr, PK_FWD))) return rc;
            LAUNCHCHK(nullptr, "dgrad_s2", launch_dgrad_s2_x3(d_dy, D.d_w, D.nt, D.wstride_cb, d_dx, nullptr, B, H, W, Cin, Cout, Cout, Cin, nullptr));
            HIPCHK(nullptr, hipDeviceSynchronize());   // before the guard frees the weights
        } else if (convt) {
            LAUNCHCHK(nullptr, "dgrad_s2", launch_dgrad_s2(d_dy, d_w, d_dx, nullptr, B, H, W, Cin, Cout, Cout, Cin, nullptr));
        } else if (Cout <= 3 && Cin % 16 == 0 && Cin <= 64) {   // the training step's path for the 64->3 and 32->2 layers
            float* d_dy4 = d.get(npix * 4);
            if (!d_dy4) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
            LAUNCHCHK(nullptr, "pad_channels", launch_pad_channels(d_dy, d_dy4, 0, (long)npix, Cout, 4, nullptr));
            LAUNCHCHK(nullptr, "dgrad_few", launch_dgrad_few(d_dy4, d_w, d_dx, nullptr, A_NONE, B, H, W, Cin, Cout, 4, Cin, nullptr));
        } else {
            ConvLayer D;
            plan_layer(D, dtype, M_CONV, Cout, Cin);
            struct Guard { ConvLayer& l; ~Guard() { free_layer(l); } } guard{D};
            if ((rc = upload_layer(nullptr, D, dtype, w, nullptr, PK_DGRAD))) return rc;
            float* d_dyp = d.get(npix * D.cin_p);
            if (!d_dyp) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
            LAUNCHCHK(nullptr, "pad_channels", launch_pad_channels(d_dy, d_dyp, 0, (long)npix, Cout, D.cin_p, nullptr));
            rc = run_conv(nullptr, dtype, nullptr, D, B, {d_dyp, H, W, d_dx, Cin, Cin, 1, nullptr, 0, A_NONE}, "op_dgrad");
            if (rc) return rc;
            HIPCHK(nullptr, hipDeviceSynchronize());   // before the guard frees the weights
        }
        HIPCHK(nullptr, hipDeviceSynchronize());
        HIPCHK(nullptr, hipMemcpy(dx, d_dx, npix * Cin * 4, hipMemcpyDeviceToHost));
    }
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(dw, d_dw, nw * 4, hipMemcpyDeviceToHost));
    if (db) HIPCHK(nullptr, hipMemcpy(db, d_db, (size_t)Cout * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}
}  // namespace

extern "C" {

int sr_op_conv3x3_bwd(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                      const float* dy, float* dx, float* dw, float* db) {
    return conv_bwd_op(device, dtype, 0, B, H, W, Cin, Cout, x, w, dy, dx, dw, db);
}

int sr_op_convT3x3s2_bwd(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                         const float* dy, float* dx, float* dw, float* db) {
    return conv_bwd_op(device, dtype, 1, B, H, W, Cin, Cout, x, w, dy, dx, dw, db);
}

int sr_op_maxpool2_bwd(int device, int B, int H, int W, int C, const float* x, const float* gy, float* gx) {
    int rc = bwd_prologue(device);
    if (rc) return rc;
    if (!x || !gy || !gx) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 1 || W < 1 || C < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    DevBufs d;
    const size_t n = (size_t)B * H * W * C, no = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    float *d_x = d.put(x, n), *d_gy = d.put(gy, no), *d_gx = d.get(n);
    if (!d_x || !d_gy || !d_gx) return fail(nullptr, SR_ERR_HIP, "hipMalloc / copy failed");
    LAUNCHCHK(nullptr, "maxpool2_bwd", launch_maxpool2_bwd(d_x, d_gy, d_gx, B, H, W, C, A_NONE, nullptr));
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(gx, d_gx, n * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_op_resize_bilinear_bwd(int device, int B, int H, int W, int C, int s, const float* gy, float* gx) {
    int rc = bwd_prologue(device);
    if (rc) return rc;
    if (!gy || !gx) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 1 || W < 1 || C < 1 || s < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    DevBufs d;
    const size_t n = (size_t)B * H * W * C;
    float *d_gy = d.put(gy, n * s * s), *d_gx = d.get(n);
    if (!d_gy || !d_gx) return fail(nullptr, SR_ERR_HIP, "hipMalloc / copy failed");
    LAUNCHCHK(nullptr, "resize_bwd", launch_resize_bwd(d_gy, d_gx, B, H, W, C, s, C, 0, nullptr, A_NONE, nullptr));
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(gx, d_gx, n * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_op_warp_bwd_flow(int device, int B, int H, int W, int C, const float* img, const float* flow, const float* g, float* gflow) {
    int rc = bwd_prologue(device);
    if (rc) return rc;
    if (!img || !flow || !g || !gflow) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 2 || W < 2 || C < 1) return fail(nullptr, SR_ERR_ARG, "dense_image_warp needs H, W >= 2");
    DevBufs d;
    const size_t npix = (size_t)B * H * W;
    float *d_i = d.put(img, npix * C), *d_f = d.put(flow, npix * 2), *d_g = d.put(g, npix * C), *d_o = d.get(npix * 2);
    if (!d_i || !d_f || !d_g || !d_o) return fail(nullptr, SR_ERR_HIP, "hipMalloc / copy failed");
    LAUNCHCHK(nullptr, "warp_bwd_flow", launch_warp_bwd_flow(d_i, d_f, d_g, d_o, B, H, W, C, 2, 0, nullptr));
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(gflow, d_o, npix * 2 * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// tecogan.train_on_batch (reference TecoGAN.py:87-108 graph, :122-131 losses, :469-479 call) up to the optimizer: the
// generator's forward, the discriminator on [est_tm1, est, next_est, warp(est_tm1), est, next_hat_est, bic_past,
// bic_pres, bic_fut] with BatchNormalization in training mode (Keras learning phase 1), the twelve weighted losses
// (the four VGG19 terms need ImageNet weights, a download: they count as 0 and their weights are ignored) and the
// backward through both networks.  Generator gradient -> the ctx's flat vector (sr_train_read_grads / sr_train_grads'
// layout), discriminator gradient -> the discriminator's (sr_disc_read_grads).
// checks shared by the host-array entry points of the combined step
static int tecogan_ready(sr_ctx* c, sr_disc* d, int B, const float* loss_weights) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (!d) return fail(c, SR_ERR_ARG, "discriminator is NULL");
    if (c->model != SR_TECOGAN) return fail(c, SR_ERR_STATE, "the combined step needs a ctx created with model SR_TECOGAN");
    if (c->pad) return fail(c, SR_ERR_STATE, "training needs a ctx whose tile size is a multiple of 8");
    if (!d->committed) return fail(c, SR_ERR_WEIGHTS, "the discriminator's weights are not committed");
    if (d->device != c->device || d->H != 4 * c->h || d->W != 4 * c->w || B > d->max_batch)
        return fail(c, SR_ERR_ARG, "discriminator (%dx%d, max batch %d, device %d) does not fit this generator", d->H, d->W, d->max_batch, d->device);
    if (!loss_weights) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    if (loss_weights[11] != 1.f) return fail(c, SR_ERR_ARG, "the warp loss weight must be 1 (reference TecoGAN.py:130); other values are not built");
    if ((rc = train_init(c))) {
        train_free(c);
        return rc;
    }
    sr_train_state* T = c->train;
    if (!T->act.count("pp") && (rc = talloc(c, "pp", (size_t)c->max_batch * c->h * c->w * 48))) return rc;
    return SR_OK;
}

// The combined step on DEVICE tensors, everything on the generator's stream (the caller has swapped the discriminator's
// stream): generator forward (taped), discriminator on [est_tm1, est, next_est, warp(est_tm1), est, next_hat_est, bic x 3]
// with batch statistics, the weighted losses, both backward passes.  d_hr / d_pp: ground truth and ping-pong target.
static int tecogan_grads_core(sr_ctx* c, sr_disc* d, int B, const float* d_lr_t, const float* d_lr_tm1, const float* d_est_tm1,
                              const float* d_next_est, const float* d_next_hat, const float* d_bic_past, const float* d_bic_pres,
                              const float* d_bic_fut, const float* d_hr, const float* const* d_feat, const float* d_pp, const float* lw,
                              float* losses) {
    sr_train_state* T = c->train;
    c->clip_active = false;
    hipStream_t st = c->stream;
    const int H = c->h, W = c->w;
    int rc;
    Tape t{c, T, B, c->prec, st};
    t.hr_warp_out = c->d_hr_warp;
    if ((rc = train_forward(t, 1, d_lr_t, d_lr_tm1, d_est_tm1, c->d_est[1], c->d_lr_warp, c->d_flow))) return rc;
    if ((rc = train_forward(t, 2, d_lr_t, d_lr_tm1, d_est_tm1, c->d_est[1], c->d_lr_warp, c->d_flow))) return rc;
    const float* est = c->d_est[1];
    // discriminator on the generated triplet: forward, losses, backward down to its 27 input channels
    const sr::Ptr9 p9 = {{d_est_tm1, est, d_next_est, c->d_hr_warp, est, d_next_hat, d_bic_past, d_bic_pres, d_bic_fut}};
    if ((rc = disc_forward(d, B, p9, true, false))) return fail(c, rc, "%s", d->err.c_str());
    LAUNCHCHK(c, "bce", launch_bce(d->logit, nullptr, 1.f, B, lw[1], d->prob, d->dlogit, d->d_loss, st));   // target D_ones (TecoGAN.py:473)
    if ((rc = disc_backward(d, B, d_feat, lw + 2, true, true))) return fail(c, rc, "%s", d->err.c_str());
    // generator backward from the weighted losses on its output + what came back through the discriminator
    const long npix = (long)B * 16 * H * W;
    // (the discriminator's reduction workspace is sized for its 64-channel HR tensor: 16x what this needs)
    // the four VGG19 terms on (hr_t, I_est_t) (TecoGAN.py:107,125,129) when their weights are attached: losses + the gradient
    // w.r.t. the estimate, added below before tanh' and the clip mask
    const float* gvgg = nullptr;
    if (c->vgg) {
        sr_vgg* v = c->vgg;
        if (!v->committed || v->device != c->device || v->H != 4 * H || v->W != 4 * W || B > v->max_batch)
            return fail(c, SR_ERR_ARG, "the attached VGG19 object (%dx%d, max batch %d, device %d, committed %d) does not fit this step", v->H,
                        v->W, v->max_batch, v->device, (int)v->committed);
        hipStream_t old = v->stream;
        v->stream = st;
        rc = vgg_loss_grads_dev(v, B, d_hr, est, lw + 6, v->g3);
        v->stream = old;
        if (rc) return fail(c, rc, "%s", v->err.c_str());
        gvgg = v->g3;
    }
    LAUNCHCHK(c, "teco_est_grad", launch_teco_est_grad(est, d_hr, d_pp, d->gS, t.a("s_out"), t.a("g_est"), d->red,
                                                       d->d_loss + 5, d->d_loss + 6, npix, lw[0], lw[10], st, gvgg));
    t.ext_g_est = true;
    t.g27 = d->gS;
    if ((rc = train_backward(t, d_lr_t, d_lr_tm1, d_est_tm1, d_hr, est, c->d_lr_warp, c->d_flow))) return rc;
    float dl[8] = {0}, gl[2] = {0}, vl[4] = {0};
    if (c->vgg) HIPCHK(c, hipMemcpyAsync(vl, c->vgg->d_loss, sizeof vl, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(dl, d->d_loss, sizeof dl, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(gl, T->d_loss, sizeof gl, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (losses) {   // Keras order: total, then the twelve outputs' unweighted losses (TecoGAN.py:105-108)
        const float parts[12] = {dl[5], dl[0], dl[1], dl[2], dl[3], dl[4], vl[0], vl[1], vl[2], vl[3], dl[6], gl[1]};
        double total = 0.0;
        for (int i = 0; i < 12; ++i) {
            losses[1 + i] = parts[i];
            if (i < 6 || i > 9 || c->vgg) total += (double)lw[i] * parts[i];
        }
        losses[0] = (float)total;
    }
    return SR_OK;
}

// runs the discriminator's kernels on the generator's stream for the lifetime of the guard
struct DiscStreamSwap {
    sr_disc* d;
    hipStream_t old;
    DiscStreamSwap(sr_disc* d_, hipStream_t st) : d(d_), old(d_->stream) { d->stream = st; }
    ~DiscStreamSwap() { d->stream = old; }
};

int sr_tecogan_set_vgg(sr_ctx* c, sr_vgg* v) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (v) {
        if (c->model != SR_TECOGAN) return fail(c, SR_ERR_STATE, "the VGG19 terms belong to the combined step of a ctx created with model SR_TECOGAN");
        if (!v->committed) return fail(c, SR_ERR_WEIGHTS, "VGG19 weights not committed (sr_vgg_set_tensor x32 + sr_vgg_commit)");
        if (v->device != c->device || v->H != 4 * c->h || v->W != 4 * c->w)
            return fail(c, SR_ERR_ARG, "VGG19 object for %dx%d on device %d does not fit this generator (%dx%d HR, device %d)", v->H, v->W, v->device,
                        4 * c->h, 4 * c->w, c->device);
    }
    c->vgg = v;
    return SR_OK;
}

int sr_tecogan_grads(sr_ctx* c, sr_disc* d, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1,
                     const float* next_est, const float* next_hat_est, const float* bic_past, const float* bic_pres,
                     const float* bic_fut, const float* hr_t, const float* const* d_targets, const float* pp_target,
                     const float* loss_weights, float* losses) {
    int rc = tecogan_ready(c, d, B, loss_weights);
    if (rc) return rc;
    if (!lr_t || !lr_tm1 || !est_tm1 || !next_est || !next_hat_est || !bic_past || !bic_pres || !bic_fut || !hr_t || !d_targets || !pp_target)
        return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    sr_train_state* T = c->train;
    hipStream_t st = c->stream;
    const size_t nlr = (size_t)B * c->h * c->w * 3 * 4, nhr = nlr * 16;
    HIPCHK(c, hipStreamSynchronize(d->stream));
    DiscStreamSwap swap_guard(d, st);
    HIPCHK(c, hipMemcpyAsync(c->d_lr_t, lr_t, nlr, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_lr_tm1, lr_tm1, nlr, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_est[0], est_tm1, nhr, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(T->act["hr"], hr_t, nhr, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(T->act["pp"], pp_target, nhr, hipMemcpyHostToDevice, st));
    const float* host_in[9] = {nullptr, nullptr, next_est, nullptr, nullptr, next_hat_est, bic_past, bic_pres, bic_fut};
    for (int i = 0; i < 9; ++i)
        if (host_in[i]) HIPCHK(c, hipMemcpyAsync(d->in9[i], host_in[i], nhr, hipMemcpyHostToDevice, st));
    const float* feat[4];
    for (int k = 0; k < 4; ++k) {
        if (!d_targets[k]) return fail(c, SR_ERR_ARG, "feature-map target %d is NULL", k);
        HIPCHK(c, hipMemcpyAsync(d->featT[k], d_targets[k], (size_t)B * (d->H >> (k + 1)) * (d->W >> (k + 1)) * DISC_COUT[k] * 4, hipMemcpyHostToDevice, st));
        feat[k] = d->featT[k];
    }
    return tecogan_grads_core(c, d, B, c->d_lr_t, c->d_lr_tm1, c->d_est[0], d->in9[2], d->in9[5], d->in9[6], d->in9[7], d->in9[8],
                              T->act["hr"], feat, T->act["pp"], loss_weights, losses);
}

static int tecogan_adam(sr_ctx* c, sr_disc* d, const float* d_gen_grads, const float* d_disc_grads, float grad_scale, float lr,
                        float beta1, float beta2, float eps, int64_t t, int train_disc);

// One whole pass of the adversarial loop body (reference TecoGAN.py:434-479) with every array uploaded ONCE and everything else
// on the device: discriminator step on the real triplets (labels 1, fresh moments, Adam step t0 + 1), on the generated ones
// (labels 0, t0 + 2), discriminator.predict on the real triplets (its feature maps become the targets, device to device),
// two combined steps (t0 + 3, t0 + 4) against the batch-flipped ground truth as ping-pong target (TecoGAN.py:476).
// Through the step-by-step entry points the same pass moves ~800 MB of host arrays (the real set three times, the feature
// maps out and in again twice); here 15 arrays go up once.
int sr_tecogan_iteration(sr_ctx* c, sr_disc* d, int B, const float* const* arrays, const float* loss_weights, float lr, float beta1,
                         float beta2, float eps, int64_t t0, int train_disc, float* losses, float* disc_losses) {
    int rc = tecogan_ready(c, d, B, loss_weights);
    if (rc) return rc;
    if (!arrays) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    for (int i = 0; i < 15; ++i)
        if (!arrays[i]) return fail(c, SR_ERR_ARG, "array %d is NULL", i);
    if (t0 < 0) return fail(c, SR_ERR_ARG, "Adam iteration must be >= 0");
    sr_train_state* T = c->train;
    hipStream_t st = c->stream;
    const size_t nlr = (size_t)B * c->h * c->w * 3 * 4, nhr = nlr * 16, img = nhr / (size_t)B;
    HIPCHK(c, hipStreamSynchronize(d->stream));
    DiscStreamSwap swap_guard(d, st);
    // arrays: 0 lr, 1 prev_lr, 2 hr, 3 prev_hr, 4 next_hr, 5 prev_hat_hr, 6 next_hat_hr, 7 est, 8 prev_est, 9 next_est,
    //         10 prev_hat_est, 11 next_hat_est, 12 bic, 13 prev_bic, 14 next_bic
    if (d->it_buf.empty()) {   // non-empty means ready: a partial failure frees what it made and leaves it empty
        std::vector<float*> bufs(13, nullptr);
        for (float*& p : bufs)
            if (hipMalloc((void**)&p, (size_t)d->max_batch * img) != hipSuccess) {
                (void)hipGetLastError();
                for (float* q : bufs)
                    if (q) (void)hipFree(q);
                return fail(c, SR_ERR_HIP, "sr_tecogan_iteration: hipMalloc of the batch buffers failed");
            }
        d->it_buf.swap(bufs);
    }
    float** hb = d->it_buf.data();          // hb[i] = arrays[i + 2]
    HIPCHK(c, hipMemcpyAsync(c->d_lr_t, arrays[0], nlr, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->d_lr_tm1, arrays[1], nlr, hipMemcpyHostToDevice, st));
    for (int i = 0; i < 13; ++i) HIPCHK(c, hipMemcpyAsync(hb[i], arrays[i + 2], nhr, hipMemcpyHostToDevice, st));
    const float *hr = hb[0], *prev_hr = hb[1], *next_hr = hb[2], *prev_hat_hr = hb[3], *next_hat_hr = hb[4], *est = hb[5], *prev_est = hb[6],
                *next_est = hb[7], *prev_hat_est = hb[8], *next_hat_est = hb[9], *bic = hb[10], *prev_bic = hb[11], *next_bic = hb[12];
    for (int b = 0; b < B; ++b)             // np.flip(hr_batch, axis=0)
        HIPCHK(c, hipMemcpyAsync((unsigned char*)T->act["pp"] + (size_t)b * img, (const unsigned char*)hr + (size_t)(B - 1 - b) * img, img,
                                 hipMemcpyDeviceToDevice, st));
    const sr::Ptr9 real = {{prev_hr, hr, next_hr, prev_hat_hr, hr, next_hat_hr, prev_bic, bic, next_bic}};
    const sr::Ptr9 fake = {{prev_est, est, next_est, prev_hat_est, est, next_hat_est, prev_bic, bic, next_bic}};
    float dl[2] = {0.f, 0.f};
    for (int k = 0; k < 2; ++k) {
        if ((rc = disc_forward(d, B, k ? fake : real, true, true))) return fail(c, rc, "%s", d->err.c_str());
        LAUNCHCHK(c, "bce", launch_bce(d->logit, nullptr, k ? 0.f : 1.f, B, 1.f, d->prob, d->dlogit, d->d_loss, st));
        if ((rc = disc_backward(d, B, nullptr, nullptr, true, false))) return fail(c, rc, "%s", d->err.c_str());
        HIPCHK(c, hipMemcpyAsync(&dl[k], d->d_loss, 4, hipMemcpyDeviceToHost, st));
        if ((rc = disc_apply(d, nullptr, 1.f, lr, beta1, beta2, eps, t0 + 1 + k, 0, k == 0))) return fail(c, rc, "%s", d->err.c_str());
    }
    if ((rc = disc_forward(d, B, real, false, false))) return fail(c, rc, "%s", d->err.c_str());
    const float* feat[4];
    for (int k = 0; k < 4; ++k) {
        HIPCHK(c, hipMemcpyAsync(d->featT[k], d->y[k], (size_t)B * (d->H >> (k + 1)) * (d->W >> (k + 1)) * DISC_COUT[k] * 4, hipMemcpyDeviceToDevice, st));
        feat[k] = d->featT[k];
    }
    for (int k = 0; k < 2; ++k) {
        if ((rc = tecogan_grads_core(c, d, B, c->d_lr_t, c->d_lr_tm1, prev_est, next_est, next_hat_est, prev_bic, bic, next_bic, hr, feat,
                                     T->act["pp"], loss_weights, losses)))
            return rc;
        if ((rc = tecogan_adam(c, d, nullptr, nullptr, 1.f, lr, beta1, beta2, eps, t0 + 3 + k, train_disc))) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    if (disc_losses) {
        disc_losses[0] = dl[0];
        disc_losses[1] = dl[1];
    }
    return SR_OK;
}

// The optimizer step of tecogan.train_on_batch: Adam (Keras 2.2.4) with iteration `t` (the reference's ONE shared Adam,
// TecoGAN.py:111: the caller keeps its counter) on the generator -- moments of tecogan's own train function, not those of
// the generator pre-training -- and, when train_disc != 0, on the discriminator too: the reference compiles tecogan while
// the discriminator is still trainable (TecoGAN.py:122 precedes :461), so its weights are among tecogan's collected
// trainable weights and move with the combined loss.
static int tecogan_adam(sr_ctx* c, sr_disc* d, const float* d_gen_grads, const float* d_disc_grads, float grad_scale, float lr,
                        float beta1, float beta2, float eps, int64_t t, int train_disc) {
    if (!c || !d) return fail(c, SR_ERR_ARG, "NULL argument");
    if (!c->train) return fail(c, SR_ERR_STATE, "no gradients: call sr_tecogan_grads first");
    if (t < 1) return fail(c, SR_ERR_ARG, "Adam iteration must be >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    sr_train_state* T = c->train;
    const size_t nb = (size_t)T->nparams * 4;
    if (!T->d_m2) {
        HIPCHK(c, hipMalloc((void**)&T->d_m2, nb));
        HIPCHK(c, hipMalloc((void**)&T->d_v2, nb));
        HIPCHK(c, hipMemsetAsync(T->d_m2, 0, nb, c->stream));
        HIPCHK(c, hipMemsetAsync(T->d_v2, 0, nb, c->stream));
    }
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    LAUNCHCHK(c, "adam", launch_adam(T->d_params, T->d_m2, T->d_v2, d_gen_grads ? d_gen_grads : T->d_grad, T->nparams, (float)lr_t, beta1,
                                     beta2, eps, d_gen_grads ? grad_scale : 1.f, c->stream));
    T->host_stale = true;
    c->mx6_valid = false;
    int rc = repack_all(c);
    if (rc) return rc;
    if (train_disc) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(d->stream));
        if ((rc = disc_apply(d, d_disc_grads, d_disc_grads ? grad_scale : 1.f, lr, beta1, beta2, eps, t, 1, false)))
            return fail(c, rc, "%s", d->err.c_str());
    }
    return SR_OK;
}

int sr_tecogan_apply(sr_ctx* c, sr_disc* d, const float* d_gen_grads, const float* d_disc_grads, float grad_scale, float lr,
                     float beta1, float beta2, float eps, int64_t t, int train_disc) {
    return tecogan_adam(c, d, d_gen_grads, d_disc_grads, grad_scale, lr, beta1, beta2, eps, t, train_disc);
}

// Both gradients of the last sr_tecogan_grads into DEVICE buffers (the all-reduce buffers of a data-parallel job).
int sr_tecogan_read_grads_dev(sr_ctx* c, sr_disc* d, float* d_gen_out, float* d_disc_out) {
    if (!c || !d) return fail(c, SR_ERR_ARG, "NULL argument");
    if (!c->train) return fail(c, SR_ERR_STATE, "no gradients: call sr_tecogan_grads first");
    HIPCHK(c, hipSetDevice(c->device));
    if (d_gen_out) HIPCHK(c, hipMemcpyAsync(d_gen_out, c->train->d_grad, (size_t)c->train->nparams * 4, hipMemcpyDeviceToDevice, c->stream));
    if (d_disc_out) HIPCHK(c, hipMemcpyAsync(d_disc_out, d->d_grad, (size_t)d->nparams * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SR_OK;
}

}  // extern "C"
