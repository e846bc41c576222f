// C-ABI host layer of libsr_hip.so (declared in include/sr_hip.h).
// Owns device memory, packs Keras-layout weights into MFMA fragment order and
// runs the generator step of reference FRVSR.py:58-66 / TecoGAN.py:68-79 as a
// fixed sequence of HIP kernel launches on one stream.  No CPU fallback.
#include "sr_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

using namespace sr;

namespace srh {

thread_local std::string g_err;
int g_conv_dbg = 0;   // ablation switches, set only by sr_bench_conv
unsigned long long* g_conv_stamps = nullptr;

static uint16_t f2bf(float f) {   // round to nearest even
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int fail(sr_ctx* c, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    g_err = tmp;
    if (c) c->err = tmp;
    return code;
}

// ---------------------------------------------------------------- weight packing
// Order inside one output-channel block: [pass][chunk][tap of pass][step][n block][hi,lo][lane] x 16 B
// (conv: one pass of 9 taps; transposed conv: the four output phases of conv_mfma.hip's pass_tap),
// lane (j = lane & 31, hh = lane >> 5) holding input channels
//   chunk*CKC + step*KS + (KS/2)*hh + e   (e < KS/2)   of output channel cb*NT*32 + n*32 + j,
// i.e. exactly the B fragment of v_mfma_f32_32x32x2_f32 (4 successive k-pairs per vector)
// or v_mfma_f32_32x32x16_bf16 that conv3x3_kernel consumes, in the order it consumes them.
static int pass_ntaps_h(int mode, int pass) { return mode == M_CONV ? 9 : (pass == 0 ? 4 : (pass == 3 ? 1 : 2)); }
static int pass_tap_h(int mode, int pass, int t) {
    static const int T[4][4] = {{0, 2, 6, 8}, {1, 7, 0, 0}, {3, 5, 0, 0}, {4, 0, 0, 0}};
    return mode == M_CONV ? t : T[pass][t];
}

// tiles: 8 x 32-pixel tiles of the largest launch this layer will see (0: unknown).  conv3x3_lds_kernel runs two persistent
// workgroups per CU; with 64 output channels per workgroup (nt = 2) a launch of few tiles rounds up badly -- the reference's
// training batch, 40 x 64 x 64, is 640 tiles on 512 slots: 64 CUs get two workgroups of two tiles, 192 get one -- so such layers
// are planned with 32 output channels per workgroup (nt = 1): twice the workgroup-tiles at ~0.6 of the cost each.  Measured on
// the training step: forward convs 2.11 -> 1.95 ms, stride-1 data gradients 1.71 -> 1.49 ms with every kernel bracketed by
// events, 8.4 -> 8.2 ms for the step as a whole.  The K order of an output channel
// does not depend on nt: results are bit-identical either way.
long conv_tiles(int B, int H, int W) { return (long)B * ((H + 7) / 8) * ((W + 31) / 32); }

void plan_layer(ConvLayer& L, int prec, int mode, int cin, int cout, long tiles) {
    L.mode = mode;
    L.cin = cin;
    L.cout = cout;
    L.cin_p = round_up(cin, 16);
    L.tapsn = false;
    const int tck = tapsn_chunk(prec);
    if (mode == M_CONV && cout <= 3 && L.cin_p % tck == 0 && L.cin_p / tck <= 2 && !getenv("SR_NO_TAPSN")) {
        L.tapsn = true;
        L.nchunk = L.cin_p / tck;
        L.steps = tck / prec_ks(prec);
        L.nt = 1;
        L.ncb = 1;
        L.wstride_cb = L.nchunk * L.steps * (prec == P_BF16X3 ? 2 : 1) * 64;
        return;
    }
    const int ks = prec_ks(prec);
    const int total_steps = L.cin_p / ks;
    int steps = prec_maxsteps(prec, mode);
    while (total_steps % steps) steps >>= 1;
    L.steps = steps;
    L.nchunk = total_steps / steps;
    L.nt = (mode == M_CONVT || cout > 32) ? 2 : 1;
    // (not for plain bf16: that ctx cannot train, and its inference measured 1.5 % slower this way)
    if (mode == M_CONV && L.nt == 2 && prec != P_BF16 && tiles > 0 && tiles * ((cout + 63) / 64) < 1280 && !getenv("SR_NO_NT1")) L.nt = 1;
    L.ncb = (cout + 32 * L.nt - 1) / (32 * L.nt);
    const int bv = prec == P_BF16X3 ? 2 : 1;
    L.wstride_cb = L.nchunk * 9 * L.steps * L.nt * bv * 64;
}

// The non-zero elements of L's packed weight buffer (see PackEnt).  kind PK_DGRAD: L is the data-gradient
// conv of a forward Conv2D (cin, cout swapped, taps mirrored) and src indexes the FORWARD layer's kernel:
//   dX[y,x,ci] = sum dY[y+1-ky, x+1-kx, co] W[ky,kx,ci,co]  =  conv3x3(dY, W'),  W'[ky,kx,co,ci] = W[2-ky,2-kx,ci,co]
std::vector<PackEnt> pack_entries(const ConvLayer& L, int prec, int kind, bool kperm) {
    const int ks = prec_ks(prec), half = ks / 2;
    const int bv = prec == P_BF16X3 ? 2 : 1;
    const int ckc = L.steps * ks;
    const int npass = L.mode == M_CONV ? 1 : 4;
    std::vector<PackEnt> out;
    // source index of the kernel element that multiplies input channel ci into output channel co at tap
    auto W = [&](int tap, int ci, int co) -> int32_t {
        if (ci >= L.cin || co >= L.cout) return -1;
        if (kind == PK_DGRAD) return (int32_t)(((size_t)(8 - tap) * L.cout + co) * L.cin + ci);   // forward layer: cin_f = L.cout, cout_f = L.cin
        if (L.mode == M_CONV) return (int32_t)(((size_t)tap * L.cin + ci) * L.cout + co);
        return (int32_t)(((size_t)tap * L.cout + co) * L.cin + ci);   // Conv2DTranspose: (kh,kw,out,in)
    };
    if (L.tapsn) {
        // [chunk][step][hi,lo][lane] x 16 B; lane (j, hh): GEMM column j = tap*3 + co (< 27), input
        // channels chunk*CK + step*KS + (KS/2)*hh + e  -- the A fragment conv3x3_tapsn_kernel keeps in registers
        const int ck = tapsn_chunk(prec);
        for (int c = 0; c < L.nchunk; ++c)
            for (int s = 0; s < L.steps; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 31, hh = lane >> 5;
                    const int tap = j / 3, co = j % 3;
                    const size_t vbase = (size_t)(c * L.steps + s) * bv * 64;
                    for (int e = 0; e < half; ++e) {
                        const int ci = c * ck + s * ks + half * hh + e;
                        const int32_t src = j < 27 ? W(tap, ci, co) : -1;
                        if (src >= 0) out.push_back({src, (uint32_t)((vbase + lane) * half + e)});
                    }
                }
        return out;
    }
    for (int cb = 0; cb < L.ncb; ++cb) {
        size_t gstep = 0;   // k-step counter in kernel consumption order
        for (int pass = 0; pass < npass; ++pass)
            for (int c = 0; c < L.nchunk; ++c)
                for (int t = 0; t < pass_ntaps_h(L.mode, pass); ++t)
                    for (int s = 0; s < L.steps; ++s, ++gstep) {
                        const int tap = pass_tap_h(L.mode, pass, t);
                        for (int n = 0; n < L.nt; ++n)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int j = lane & 31, hh = lane >> 5;
                                const int co = (cb * L.nt + n) * 32 + j;
                                const size_t vbase = (size_t)cb * L.wstride_cb + (gstep * L.nt + n) * bv * 64;
                                for (int e = 0; e < half; ++e) {
                                    // kperm: the input is split-planar, k index (hh, e) of a 16-channel step is channel split_perm(hh, e)
                                    const int ci = c * ckc + s * ks + (kperm ? split_perm(hh, e) : half * hh + e);
                                    const int32_t src = W(tap, ci, co);
                                    if (src >= 0) out.push_back({src, (uint32_t)((vbase + lane) * half + e)});
                                }
                            }
                    }
    }
    return out;
}

// conv_trunk.hip: [half of cout][16-channel plane of cin][tap][hi, lo][lane (j, hh)] x 8 bf16, lane (j, hh) element e =
// kernel[tap][16 plane + split_perm(hh, e)][32 half + j]; hi / lo blocks are 512 elements apart like everywhere else
std::vector<PackEnt> trunk_entries(int cin, int cout) {
    std::vector<PackEnt> out;
    for (int hf = 0; hf < 2; ++hf)
        for (int c = 0; c < 4; ++c)
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 31, hh = lane >> 5, co = 32 * hf + j;
                    const size_t vbase = ((size_t)((hf * 4 + c) * 9 + t) * 2) * 64;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * c + split_perm(hh, e);
                        if (ci < cin && co < cout)
                            out.push_back({(int32_t)(((size_t)t * cin + ci) * cout + co), (uint32_t)((vbase + lane) * 8 + e)});
                    }
                }
    return out;
}

// conv_trunk.hip trunk_convT_kernel (conv2d_transpose_1): the same fragment layout from a Conv2DTranspose kernel (kh, kw, out, in)
std::vector<PackEnt> trunkT_entries(int cin, int cout) {
    std::vector<PackEnt> out;
    for (int hf = 0; hf < 2; ++hf)
        for (int c = 0; c < 4; ++c)
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 31, hh = lane >> 5, co = 32 * hf + j;
                    const size_t vbase = ((size_t)((hf * 4 + c) * 9 + t) * 2) * 64;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * c + split_perm(hh, e);
                        if (ci < cin && co < cout)
                            out.push_back({(int32_t)(((size_t)t * cout + co) * cin + ci), (uint32_t)((vbase + lane) * 8 + e)});
                    }
                }
    return out;
}

// conv_fnet.hip: [32-channel block of cout][16-channel plane of cin][tap][hi, lo][lane (j, hh)] x 8 bf16, the same fragment as
// trunk_entries for any (cin, cout)
std::vector<PackEnt> fconv_entries(int cin, int cout) {
    std::vector<PackEnt> out;
    const int ncb = (cout + 31) / 32, npl = (cin + 15) / 16;
    for (int cb = 0; cb < ncb; ++cb)
        for (int c = 0; c < npl; ++c)
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 31, hh = lane >> 5, co = 32 * cb + j;
                    const size_t vbase = ((size_t)((cb * npl + c) * 9 + t) * 2) * 64;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * c + split_perm(hh, e);
                        if (ci < cin && co < cout)
                            out.push_back({(int32_t)(((size_t)t * cin + ci) * cout + co), (uint32_t)((vbase + lane) * 8 + e)});
                    }
                }
    return out;
}

// host side of the packing: zero buffer + the listed elements in the arithmetic's storage type
static std::vector<uint8_t> pack_apply(const std::vector<PackEnt>& ents, size_t bytes, int prec, const float* w) {
    std::vector<uint8_t> out(bytes, 0);
    for (const PackEnt& p : ents) {
        const float v = w[p.src];
        if (prec == P_FP32) {
            memcpy(&out[(size_t)p.dst * 4], &v, 4);
        } else {
            const uint16_t hi = f2bf(v);
            memcpy(&out[(size_t)p.dst * 2], &hi, 2);
            if (prec == P_BF16X3) {
                const uint16_t lo = f2bf(v - bf2f(hi));
                memcpy(&out[((size_t)p.dst + 512) * 2], &lo, 2);
            }
        }
    }
    return out;
}

int upload_layer(sr_ctx* c, ConvLayer& L, int prec, const float* w, const float* bias, int kind) {
    L.ents = pack_entries(L, prec, kind);
    std::vector<uint8_t> packed = pack_apply(L.ents, (size_t)L.ncb * L.wstride_cb * 16, prec, w);
    std::vector<float> b((size_t)L.ncb * L.nt * 32, 0.f);
    if (bias)
        for (int i = 0; i < L.cout; ++i) b[i] = bias[i];
    if (L.d_w) (void)hipFree(L.d_w);
    if (L.d_bias) (void)hipFree(L.d_bias);
    L.d_w = nullptr;
    L.d_bias = nullptr;
    HIPCHK(c, hipMalloc(&L.d_w, packed.size()));
    HIPCHK(c, hipMalloc((void**)&L.d_bias, b.size() * 4));
    HIPCHK(c, hipMemcpy(L.d_w, packed.data(), packed.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(L.d_bias, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    L.wbytes = (double)packed.size();
    return SR_OK;
}

int upload_alt(sr_ctx* c, ConvLayer& L, size_t bytes, const float* w) {
    std::vector<uint8_t> packed = pack_apply(L.ents2, bytes, P_BF16X3, w);
    if (L.d_w2) (void)hipFree(L.d_w2);
    L.d_w2 = nullptr;
    HIPCHK(c, hipMalloc(&L.d_w2, packed.size()));
    HIPCHK(c, hipMemcpy(L.d_w2, packed.data(), packed.size(), hipMemcpyHostToDevice));
    return SR_OK;
}

void free_layer(ConvLayer& L) {
    void* ps[] = {L.d_w, L.d_bias, L.d_proj, L.d_ents, L.d_proj_ents, L.d_w2, L.d_ents2, L.d_w3, L.d_ents3, L.d_w6};
    for (void* p : ps)
        if (p) (void)hipFree(p);
    L.d_w = nullptr;
    L.d_bias = nullptr;
    L.d_proj = nullptr;
    L.d_ents = nullptr;
    L.d_proj_ents = nullptr;
    L.d_w2 = nullptr;
    L.d_ents2 = nullptr;
    L.d_w3 = nullptr;
    L.d_ents3 = nullptr;
    L.d_w6 = nullptr;
}

// Tap-projection weights for conv_mfma.hip epilogue_proj: the following conv's kernel w[3,3,64,cout<=3]
// as a 64 x 32 matrix (column nn = tap*3 + co, 27 used), packed as MFMA A fragments whose K order is
// the accumulator layout of the transposed conv: k-step (n, s), lane (nn, hh), element e holds channel
//   n*32 + 16*s + 8*(e>>2) + 4*hh + (e&3).
// [k-step][hi,lo][lane] x 16 B for bf16x3, [k-step][lane] for bf16, [k-step][e>>2][lane] for fp32.
size_t proj_bytes(int prec) { return (size_t)4 * (prec == P_BF16 ? 1 : 2) * 64 * 16; }

std::vector<PackEnt> proj_entries(int prec, int cout) {
    const int vps = prec == P_BF16 ? 1 : 2;
    std::vector<PackEnt> out;
    for (int n = 0; n < 2; ++n)
        for (int s = 0; s < 2; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int nn = lane & 31, hh = lane >> 5, tap = nn / 3, co = nn % 3;
                const size_t vbase = (size_t)(n * 2 + s) * vps * 64;
                if (nn >= 27 || co >= cout) continue;
                for (int e = 0; e < 8; ++e) {
                    const int ch = n * 32 + 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3);
                    const int32_t src = (int32_t)(((size_t)tap * 64 + ch) * cout + co);
                    const uint32_t dst = prec == P_FP32 ? (uint32_t)((vbase + (e >> 2) * 64 + lane) * 4 + (e & 3))
                                                        : (uint32_t)((vbase + lane) * 8 + e);
                    out.push_back({src, dst});
                }
            }
    return out;
}

int upload_proj(sr_ctx* c, ConvLayer& L, int prec, const float* w, int cout) {
    L.proj_ents = proj_entries(prec, cout);
    std::vector<uint8_t> out = pack_apply(L.proj_ents, proj_bytes(prec), prec, w);
    if (L.d_proj) (void)hipFree(L.d_proj);
    L.d_proj = nullptr;
    HIPCHK(c, hipMalloc(&L.d_proj, out.size()));
    HIPCHK(c, hipMemcpy(L.d_proj, out.data(), out.size(), hipMemcpyHostToDevice));
    return SR_OK;
}

// ---------------------------------------------------------------- profiling helpers
ProfEntry* prof_entry(sr_ctx* c, const char* name) {
    for (auto& e : c->prof)
        if (e.name == name) return &e;
    c->prof.push_back(ProfEntry());
    c->prof.back().name = name;
    return &c->prof.back();
}

hipEvent_t get_event(sr_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_resolve(sr_ctx* c) {
    for (auto& p : c->prof) {
        for (auto& ev : p.pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(ev.second);
            (void)hipEventElapsedTime(&ms, ev.first, ev.second);
            p.ms += ms;
            c->ev_pool.push_back(ev.first);
            c->ev_pool.push_back(ev.second);
        }
        p.pending.clear();
    }
}

// ---------------------------------------------------------------- layer launch

int run_conv(sr_ctx* c, int prec, hipStream_t st, const ConvLayer& L, int B, const ConvIO& io, const char* name) {
    ConvArgs a = {};
    a.in = io.in;
    a.wpack = L.d_w;
    a.bias = L.d_bias;
    a.out = io.out;
    a.res = io.res;
    a.B = B;
    a.H = io.H;
    a.W = io.W;
    a.cin_p = L.cin_p;
    a.out_cs = io.out_cs;
    a.res_cs = io.res_cs;
    a.cout_store = io.cout_store;
    a.out_f32 = io.out_f32;
    a.act = io.act;
    a.in_planar = io.in_planar;
    a.out_planar = io.out_planar;
    a.proj_w = io.proj_w;
    a.res_mask = io.res_mask;
    a.in_split = io.in_split;
    a.pool_out = io.pool_out;
    a.gather_out = io.gather_out;
    a.gather_bias = io.gather_bias;
    a.gather_cs = io.gather_cs;
    a.gather_cout = io.gather_cout;
    a.gather_act = io.gather_act;
    if (io.gather_out && (!io.proj_w || !io.gather_bias || io.gather_cout < 1 || io.gather_cout > 3 || io.gather_cs < io.gather_cout))
        return fail(c, SR_ERR_ARG, "%s: the in-kernel tap gather needs the tap projection and a bias of <= 3 channels", name);
    if (io.pool_out && (L.tapsn || L.mode != M_CONV || io.res || io.out_planar || io.res_mask || (io.H & 1) || (io.W & 1) ||
                        (io.cout_store * (io.out_f32 ? 4 : 2)) % 16 || (io.out_cs * (io.out_f32 ? 4 : 2)) % 16))
        return fail(c, SR_ERR_ARG, "%s: the fused max pooling needs a plain 3x3 conv on an even-sized image with vector-aligned channels", name);
    if (io.in_split) {
        if (L.mode != M_CONVT || prec != P_BF16X3 || !io.in_planar || !L.d_w2 || L.nchunk != 1 || L.steps != 4)
            return fail(c, SR_ERR_ARG, "%s: a split-planar input needs the bf16x3 64-channel transposed conv with its second packing", name);
        a.wpack = L.d_w2;
    }
    if (io.res_mask && (L.tapsn || L.mode != M_CONV || prec == P_BF16 || !io.res || io.out_planar || io.proj_w ||
                        (io.act != A_RELU && io.act != A_LRELU) || io.cout_store % 8 || io.out_cs % 8 || io.res_cs % 8))
        return fail(c, SR_ERR_ARG, "%s: masked data-gradient needs an fp32/bf16x3 3x3 conv with vector-aligned strides", name);
    if (io.proj_w && (L.mode != M_CONVT || L.cout != 64 || L.nt != 2 || !io.out_f32 || io.res))
        return fail(c, SR_ERR_ARG, "%s: tap projection needs a 64-channel transposed conv", name);
    if (io.in_planar && (L.tapsn || L.steps != prec_maxsteps(prec, L.mode) || (L.mode == M_CONVT && L.nchunk != 1)))
        return fail(c, SR_ERR_ARG, "%s: planar input needs 64-byte chunks", name);
    if (io.out_planar && (L.tapsn || L.mode != M_CONV || (io.cout_store * (io.out_f32 ? 4 : 2)) % 64 || io.cout_store != io.out_cs ||
                          (io.res && io.res_cs != io.out_cs) || io.act == A_TANH))
        return fail(c, SR_ERR_ARG, "%s: planar output needs whole 64-byte planes", name);
    const int th = L.mode == M_CONV ? 8 : 4;
    a.tiles_x = (io.W + TW - 1) / TW;
    a.tiles_y = (io.H + th - 1) / th;
    a.nchunk = L.nchunk;
    a.wstride_cb = L.wstride_cb;
    a.dbg = g_conv_dbg;
    a.stamps = g_conv_stamps;
    if (io.cout_store > L.ncb * L.nt * 32) return fail(c, SR_ERR_ARG, "%s: cout_store out of range", name);
    const double px = (double)B * io.H * io.W;
    const double esz_in = prec == P_BF16 ? 2 : 4, esz_out = io.out_f32 ? 4 : 2;
    const double opx = L.mode == M_CONV ? px : 4 * px;
    const double flops = px * 9.0 * L.cin * L.cout * 2.0;
    // in-kernel gather: the image itself + the partial sums of the 8 x 64 tiles' border rings (written, then read by the fix kernel)
    const double out_px_bytes = io.gather_out ? io.gather_cout * 4.0 + 2.0 * 16.0 * (140 + 148) / 512.0
                                              : io.cout_store * esz_out * (io.res ? 2 : 1);
    const double bytes = px * L.cin * esz_in + opx * out_px_bytes + L.wbytes;
    hipError_t e;
    if (L.tapsn && (!io.out_f32 || io.cout_store > 3)) return fail(c, SR_ERR_ARG, "%s: taps-in-N conv needs an fp32 output of <= 3 channels", name);
    auto launch = [&]() {
        if (L.tapsn) return launch_conv_tapsn(prec, L.nchunk, a, st);
        hipError_t le = launch_conv(prec, L.mode, L.nt, L.steps, a, L.ncb, st);
        if (le == hipSuccess && a.gather_out) le = launch_gather_fix(a, st);   // the tile-border pixels
        return le;
    };
    if (c) {
        ProfScope ps(c, name, flops, bytes);
        e = launch();
    } else {
        e = launch();
    }
    if (e != hipSuccess) return fail(c, SR_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e));
    return SR_OK;
}


// every buffer is followed by 256 zero bytes (conv_trunk.hip reads the halo pixels that lie outside the image from there)
// and a 2 KiB dump area (where its lanes outside the image load and store); 4 KiB in all
int alloc_buf(sr_ctx* c, const char* name, size_t bytes) {
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, bytes + 4096));
    HIPCHK(c, hipMemset((unsigned char*)p + bytes, 0, 4096));
    c->buf[name] = p;
    c->buf_img_bytes[name] = bytes / (size_t)c->max_batch;
    return SR_OK;
}

// split-planar trunk (conv_trunk.hip) for this step?
// The split-planar kernels address their tensors -- and the zero / dump tail behind image max_batch - 1 -- with 32-bit byte
// offsets from the lane's first image, so the largest of them ("up1": 1 KiB per LR pixel) bounds the ctx: beyond
// max_batch * h * w * 1024 + 4 KiB = 4 GiB (max_batch 71 at 184x320) every step takes the generic conv_mfma path instead
// (64-bit addressing), whatever B is.
bool use_split(const sr_ctx* c, int B) {
    if (c->prec != P_BF16X3 || !c->split_trunk || !c->srnet[0].d_w2) return false;
    if ((double)c->max_batch * c->h * c->w * 1024.0 + 4096.0 > 4294967295.0) return false;
    return c->split_trunk >= 2 || (size_t)B * c->h * c->w > 65536;
}

// stream and activation buffers of the lane being enqueued (whole batch on `stream` outside a fork)
hipStream_t lane_st(const sr_ctx* c) { return c->lane_stream ? c->lane_stream : c->stream; }
void* lane_buf(sr_ctx* c, const char* name) {
    return (unsigned char*)c->buf[name] + (size_t)c->lane_b0 * c->buf_img_bytes[name];
}

// expected (cin, cout, mode) of every weighted layer, reference layers.py:30-44,67-75
extern const int FNET_SPEC[14][2] = {{6, 32}, {32, 32}, {32, 64}, {64, 64}, {64, 128}, {128, 128}, {128, 256},
                                     {256, 256}, {256, 128}, {128, 128}, {128, 64}, {64, 64}, {64, 32}, {32, 2}};

void srnet_spec(int i, int& cin, int& cout, int& mode) {
    mode = (i == 21 || i == 22) ? M_CONVT : M_CONV;
    cin = i == 0 ? 51 : 64;
    cout = i == 23 ? 3 : 64;
}

int check_ready(sr_ctx* c, int B) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->committed) return fail(c, SR_ERR_WEIGHTS, "weights not committed (sr_load_tensor x76 + sr_commit_weights)");
    if (B < 1 || B > c->max_batch) return fail(c, SR_ERR_ARG, "batch %d outside [1, max_batch=%d]", B, c->max_batch);
    HIPCHK(c, hipSetDevice(c->device));
    return SR_OK;
}

// split-planar FNet (conv_fnet.hip) for this step?  Same rule and the same 32-bit limit as the trunk (use_split): FNet's largest
// tensor ("u0", 256 B per LR pixel) is a quarter of the trunk's.
bool use_fnet_split(const sr_ctx* c, int B) {
    if (c->prec != P_BF16X3 || !c->fnet_split || !c->fnet[0].d_w2) return false;
    if ((double)c->max_batch * c->h * c->w * 256.0 + 4096.0 > 4294967295.0) return false;
    return c->fnet_split >= 2 || (size_t)B * c->h * c->w > 65536;
}

// FNet on conv_fnet.hip: every activation split-planar, MaxPooling2D inside the producing conv, conv2d_13 writes the fp32
// NHWC tensor conv2d_14's taps-in-N kernel reads.  Buffers are the ones run_fnet uses (same bytes per element).
static int run_fnet_split(sr_ctx* c, int B, const float* d_lr_t, const float* d_lr_tm1, float* d_flow) {
    hipStream_t st = lane_st(c);
    const int H = c->h, W = c->w;
    auto b = [&](const char* n) { return (unsigned char*)lane_buf(c, n); };
    // byte offset of the zero / dump tail (behind image max_batch - 1) from the lane's first image
    auto tail = [&](const char* n) { return (unsigned)((size_t)(c->max_batch - c->lane_b0) * c->buf_img_bytes[n]); };
    {
        ProfScope ps(c, "fnet.input", 0, (double)B * H * W * (24 + 64));
        LAUNCHCHK(c, "fnet.input", launch_fnet_input_split(d_lr_t, d_lr_tm1, b("x0"), B, H, W, st));
    }
    char nm[32];
    int k = 0;
    auto conv = [&](const char* in, int h, int w, const char* out, bool pool, bool f32) -> int {
        const ConvLayer& L = c->fnet[k];
        snprintf(nm, sizeof nm, "fnet.conv%d", k + 1);
        ++k;
        FConvArgs fa = {};
        fa.in = b(in);
        fa.out = b(out);
        fa.wpack = (const unsigned char*)L.d_w2;
        fa.bias = L.d_bias;
        fa.zero_off = tail(in);
        fa.dump_off = tail(out);
        fa.B = B;
        fa.H = h;
        fa.W = w;
        fa.npl = L.cin_p / 16;
        fa.ncb = (L.cout + 31) / 32;
        fa.pool = pool;
        fa.out_f32 = f32;
        const double px = (double)B * h * w;
        ProfScope ps(c, nm, px * 9.0 * L.cin * L.cout * 2.0, px * 4.0 * (L.cin_p + (pool ? L.cout / 4.0 : L.cout)) + (double)fnet_weight_bytes(L.cin, L.cout));
        LAUNCHCHK(c, nm, launch_fnet_conv(fa, st));
        return SR_OK;
    };
    auto resize = [&](const char* in, int h, int w, int ch, const char* out, int idx) -> int {
        snprintf(nm, sizeof nm, "fnet.resize%d", idx);
        ProfScope ps(c, nm, 0, (double)B * h * w * ch * 4 * 5.0);
        LAUNCHCHK(c, nm, launch_resize2x_split(b(in), b(out), B, h, w, ch, st));
        return SR_OK;
    };
    int rc;
    // encoder (layers.py:30-36): conv, conv + pool at 32 / 64 / 128 channels, then 256 channels at 1/8 size
    if ((rc = conv("x0", H, W, "a0", false, false))) return rc;
    if ((rc = conv("a0", H, W, "p1", true, false))) return rc;
    if ((rc = conv("p1", H / 2, W / 2, "a1", false, false))) return rc;
    if ((rc = conv("a1", H / 2, W / 2, "p2", true, false))) return rc;
    if ((rc = conv("p2", H / 4, W / 4, "a2", false, false))) return rc;
    if ((rc = conv("a2", H / 4, W / 4, "p3", true, false))) return rc;
    if ((rc = conv("p3", H / 8, W / 8, "a3", false, false))) return rc;
    if ((rc = conv("a3", H / 8, W / 8, "b3", false, false))) return rc;
    // decoder (layers.py:37-43)
    if ((rc = resize("b3", H / 8, W / 8, 256, "u2", 1))) return rc;
    if ((rc = conv("u2", H / 4, W / 4, "a2", false, false))) return rc;
    if ((rc = conv("a2", H / 4, W / 4, "b2", false, false))) return rc;
    if ((rc = resize("b2", H / 4, W / 4, 128, "u1", 2))) return rc;
    if ((rc = conv("u1", H / 2, W / 2, "a1", false, false))) return rc;
    if ((rc = conv("a1", H / 2, W / 2, "b1", false, false))) return rc;
    if ((rc = resize("b1", H / 2, W / 2, 64, "u0", 3))) return rc;
    if ((rc = conv("u0", H, W, "a0", false, true))) return rc;          // conv2d_13: fp32 NHWC for the taps-in-N kernel
    return run_conv(c, c->prec, st, c->fnet[13], B, {b("a0"), H, W, d_flow, 2, 2, 1, nullptr, 0, A_TANH}, "fnet.conv14");
}

// FNet: reference layers.py:27-45
int run_fnet(sr_ctx* c, int B, const float* d_lr_t, const float* d_lr_tm1, float* d_flow) {
    if (c->step_fnet_split) return run_fnet_split(c, B, d_lr_t, d_lr_tm1, d_flow);
    const int bf = c->prec == P_BF16, P = c->prec;
    const int af = bf ? 0 : 1;   // activations are float?
    hipStream_t st = lane_st(c);
    const int H = c->h, W = c->w;
    auto b = [&](const char* n) { return lane_buf(c, n); };
    {
        ProfScope ps(c, "fnet.input", 0, (double)B * H * W * (24 + 16 * (bf ? 2 : 4)));
        LAUNCHCHK(c, "fnet.input", launch_fnet_input(d_lr_t, d_lr_tm1, b("x0"), bf, B, H, W, st));
    }
    struct Lvl { int H, W; } lv[4] = {{H, W}, {H / 2, W / 2}, {H / 4, W / 4}, {H / 8, W / 8}};
    const char* A[4] = {"a0", "a1", "a2", "a3"};
    const char* Bb[4] = {"b0", "b1", "b2", "b3"};
    const char* Pn[4] = {nullptr, "p1", "p2", "p3"};
    const char* Un[3] = {"u0", "u1", "u2"};
    const int ch[4] = {32, 64, 128, 256};
    char nm[32];
    int k = 0;
    const void* x = b("x0");
    // encoder
    for (int l = 0; l < 4; ++l) {
        snprintf(nm, sizeof nm, "fnet.conv%d", k + 1);
        int rc = run_conv(c, P, st, c->fnet[k++], B, {x, lv[l].H, lv[l].W, b(A[l]), ch[l], ch[l], af, nullptr, 0, A_LRELU}, nm);
        if (rc) return rc;
        snprintf(nm, sizeof nm, "fnet.conv%d", k + 1);
        ConvIO io2 = {b(A[l]), lv[l].H, lv[l].W, b(Bb[l]), ch[l], ch[l], af, nullptr, 0, A_LRELU};
        // MaxPooling2D in the producing conv's epilogue (the unpooled tensor has no other reader); SR_NO_FUSE_POOL=1: separate kernel
        const bool fuse_pool = l < 3 && c->fuse_pool;
        if (fuse_pool) io2.pool_out = b(Pn[l + 1]);
        rc = run_conv(c, P, st, c->fnet[k++], B, io2, nm);
        if (rc) return rc;
        x = b(Bb[l]);
        if (fuse_pool) {
            x = b(Pn[l + 1]);
        } else if (l < 3) {
            snprintf(nm, sizeof nm, "fnet.pool%d", l + 1);
            const double bytes = (double)B * lv[l].H * lv[l].W * ch[l] * (bf ? 2 : 4) * 1.25;
            ProfScope ps(c, nm, 0, bytes);
            LAUNCHCHK(c, nm, launch_maxpool2(x, b(Pn[l + 1]), bf, B, lv[l].H, lv[l].W, ch[l], st));
            x = b(Pn[l + 1]);
        }
    }
    // decoder
    for (int l = 2; l >= 0; --l) {
        snprintf(nm, sizeof nm, "fnet.resize%d", 3 - l);
        {
            const double bytes = (double)B * lv[l + 1].H * lv[l + 1].W * ch[l + 1] * (bf ? 2 : 4) * 5.0;
            ProfScope ps(c, nm, 0, bytes);
            LAUNCHCHK(c, nm, launch_resize2x(x, b(Un[l]), bf, B, lv[l + 1].H, lv[l + 1].W, ch[l + 1], st));
        }
        x = b(Un[l]);
        if (l > 0) {
            snprintf(nm, sizeof nm, "fnet.conv%d", k + 1);
            int rc = run_conv(c, P, st, c->fnet[k++], B, {x, lv[l].H, lv[l].W, b(A[l]), ch[l], ch[l], af, nullptr, 0, A_LRELU}, nm);
            if (rc) return rc;
            snprintf(nm, sizeof nm, "fnet.conv%d", k + 1);
            rc = run_conv(c, P, st, c->fnet[k++], B, {b(A[l]), lv[l].H, lv[l].W, b(Bb[l]), ch[l], ch[l], af, nullptr, 0, A_LRELU}, nm);
            if (rc) return rc;
            x = b(Bb[l]);
        }
    }
    int rc = run_conv(c, P, st, c->fnet[12], B, {x, H, W, b("a0"), 32, 32, af, nullptr, 0, A_LRELU}, "fnet.conv13");
    if (rc) return rc;
    return run_conv(c, P, st, c->fnet[13], B, {b("a0"), H, W, d_flow, 2, 2, 1, nullptr, 0, A_TANH}, "fnet.conv14");
}

// SRNet: reference layers.py:66-76; input already concatenated into buf "s_in".
// The LR-resolution trunk (head + 10 residual blocks) runs over sub-batches of `trunk_sub` clips so
// that its two ping-pong activation tensors stay resident in the 256 MiB Infinity Cache between
// layers instead of streaming through HBM 41 times per frame (SR_TRUNK_SUB=n overrides the automatic
// choice; a value >= the batch disables the split).
int run_srnet(sr_ctx* c, int B, float* d_out, bool split) {
    const int P = c->prec, af = c->prec == P_BF16 ? 0 : 1;
    hipStream_t st = lane_st(c);
    const int H = c->h, W = c->w;
    const size_t es = af ? 4 : 2;
    // trunk sub-batch: the two ping-pong tensors of every lane in flight should stay inside the 256 MiB
    // Infinity Cache with room for the rest of the traffic (measured: +3.5 % at 2 lanes x 2 clips)
    int sub = c->prof_on ? B : c->trunk_sub;   // per-kernel timing: whole-batch launches, one lane
    if (sub <= 0) {
        const size_t img = (size_t)H * W * 64 * es;
        const int lanes = c->lane_stream ? c->dual + 1 : 1;
        sub = (int)std::max<size_t>(1, ((size_t)128 << 20) / ((size_t)lanes * 2 * img));
        if (split) sub = 4;   // conv_trunk.hip: four clips (x + y = 120 MB at 184x320) stay in the Infinity Cache and still give every
                              // workgroup pair several tiles (measured on one box: 2202 frames/s, 2141 with the whole lane, 2120 with 2)
    }
    sub = std::min(sub, B);
    for (int b0 = 0; b0 < B; b0 += sub) {
        const int nb = b0 + sub <= B ? sub : B - b0;
        const size_t off = (size_t)b0 * H * W * 64 * es;
        unsigned char* s_in = (unsigned char*)lane_buf(c, "s_in") + off;
        unsigned char* sx = (unsigned char*)lane_buf(c, "sx") + off;
        unsigned char* sy = (unsigned char*)lane_buf(c, "sy") + off;
        if (split) {
            // all three tensors are split-planar and of one size; the zero tail sits behind image max_batch - 1
            const size_t img = (size_t)H * W * 256;
            const size_t tail = (size_t)(c->max_batch - c->lane_b0 - b0) * img;
            auto trunk = [&](const ConvLayer& L, const unsigned char* in, unsigned char* out, const unsigned char* res, int relu,
                             const char* name) -> int {
                TrunkArgs ta = {in, out, res, (const unsigned char*)L.d_w2, L.d_bias, (unsigned)tail, 0u, nb, H, W, relu, 0, 0, 0};
                const double px = (double)nb * H * W;
                ProfScope ps(c, name, px * 9.0 * L.cin * L.cout * 2.0, px * 256.0 * (res ? 3 : 2) + (double)trunk_weight_bytes());
                if (tail + 4096 > 4294967295.0) return fail(c, SR_ERR_ARG, "%s: tensor too large for 32-bit offsets", name);
                LAUNCHCHK(c, name, launch_trunk_conv(ta, st));
                return SR_OK;
            };
            if (c->trunk_mx6 && c->mx6_valid) {
                // head (bf16x3 arithmetic) -> sx in mx6; the blocks ping-pong sx / sy in mx6
                auto trunk6 = [&](const ConvLayer& L, const unsigned char* in, unsigned char* out, const unsigned char* res, int relu,
                                  int out_split, const char* name) -> int {
                    TrunkArgs ta = {in, out, res, (const unsigned char*)L.d_w6, L.d_bias, (unsigned)tail, 0u, nb, H, W, relu, 0, 0, 0};
                    const double px = (double)nb * H * W;
                    ProfScope ps(c, name, px * 9.0 * L.cin * L.cout * 2.0, px * 256.0 * (res ? 3 : 2) + (double)trunk6_weight_bytes());
                    if (tail + 4096 > 4294967295.0) return fail(c, SR_ERR_ARG, "%s: tensor too large for 32-bit offsets", name);
                    LAUNCHCHK(c, name, launch_trunk6_conv(ta, out_split, st));
                    return SR_OK;
                };
                int rc;
                {   // the head conv (bf16x3 arithmetic on the split-planar input) writes the format itself
                    const ConvLayer& L = c->srnet[0];
                    TrunkArgs ta = {s_in, sx, nullptr, (const unsigned char*)L.d_w2, L.d_bias, (unsigned)tail, 0u, nb, H, W, 1, 0, 0, 0, 1};
                    const double px = (double)nb * H * W;
                    ProfScope ps(c, "srnet.head", px * 9.0 * L.cin * L.cout * 2.0, px * 256.0 * 2 + (double)trunk_weight_bytes());
                    if (tail + 4096 > 4294967295.0) return fail(c, SR_ERR_ARG, "srnet.head: tensor too large for 32-bit offsets");
                    LAUNCHCHK(c, "srnet.head", launch_trunk_conv(ta, st));
                }
                for (int r = 0; r < 10; ++r) {
                    if ((rc = trunk6(c->srnet[1 + 2 * r], sx, sy, nullptr, 1, 0, "srnet.res"))) return rc;
                    // x = x + conv(y) in place; the last one writes split-planar records into s_in for conv2d_transpose_1 (not in
                    // place: its output planes are other workgroups' residual planes)
                    if ((rc = trunk6(c->srnet[2 + 2 * r], sy, r == 9 ? s_in : sx, sx, 0, r == 9, "srnet.res"))) return rc;
                }
                continue;
            }
            int rc = trunk(c->srnet[0], s_in, sx, nullptr, 1, "srnet.head");
            if (rc) return rc;
            for (int r = 0; r < 10; ++r) {
                if ((rc = trunk(c->srnet[1 + 2 * r], sx, sy, nullptr, 1, "srnet.res"))) return rc;
                if ((rc = trunk(c->srnet[2 + 2 * r], sy, sx, sx, 0, "srnet.res"))) return rc;   // x = x + conv(y), in place
            }
            continue;
        }
        const int pl = c->planar_trunk;   // trunk activations in 64-byte planes (kernels.h)
        int rc = run_conv(c, P, st, c->srnet[0], nb, {s_in, H, W, sx, 64, 64, af, nullptr, 0, A_RELU, 0, pl}, "srnet.head");
        if (rc) return rc;
        for (int r = 0; r < 10; ++r) {
            rc = run_conv(c, P, st, c->srnet[1 + 2 * r], nb, {sx, H, W, sy, 64, 64, af, nullptr, 0, A_RELU, pl, pl}, "srnet.res");
            if (rc) return rc;
            // x = x + conv(y): each thread reads then writes its own element, in place is safe
            rc = run_conv(c, P, st, c->srnet[2 + 2 * r], nb, {sy, H, W, sx, 64, 64, af, sx, 64, A_NONE, pl, pl}, "srnet.res");
            if (rc) return rc;
        }
    }
    auto b = [&](const char* n) { return lane_buf(c, n); };
    // split && trunk_up1: conv2d_transpose_1 on conv_trunk.hip's transposed-conv kernel, split-planar in AND out (the input
    // staging of conv2d_transpose_2 is then a plain copy); SR_NO_TRUNK_UP1=1: the generic transposed-conv kernel, fp32 NHWC out
    const bool up1_split = split && c->trunk_up1;
    const char* trunk_out = split && c->trunk_mx6 && c->mx6_valid ? "s_in" : "sx";   // where the loop above left the trunk's output
    int rc;
    if (up1_split) {
        const ConvLayer& L = c->srnet[21];
        const size_t left = (size_t)(c->max_batch - c->lane_b0);
        const size_t tail_in = left * H * W * 256, tail_out = 4 * tail_in;
        if (tail_out + 4096 > 4294967295.0) return fail(c, SR_ERR_ARG, "srnet.up1: tensor too large for 32-bit offsets");
        TrunkArgs ta = {(const unsigned char*)b(trunk_out), (unsigned char*)b("up1"), nullptr, (const unsigned char*)L.d_w2, L.d_bias,
                        (unsigned)tail_in, (unsigned)tail_out, B, H, W, 1, 0, 0, 0};
        const double px = (double)B * H * W;
        ProfScope ps(c, "srnet.up1", px * 9.0 * 64 * 64 * 2.0, px * 256.0 * 5 + (double)trunk_weight_bytes());
        LAUNCHCHK(c, "srnet.up1", launch_trunk_convT(ta, st));
    } else {
        ConvIO up1 = {b(trunk_out), H, W, b("up1"), 64, 64, af, nullptr, 0, A_RELU, split ? 1 : c->planar_trunk, 0};
        up1.in_split = split ? 1 : 0;
        rc = run_conv(c, P, st, c->srnet[21], B, up1, "srnet.up1");
        if (rc) return rc;
    }
    const int u2p = up1_split ? 1 : 0;   // conv2d_transpose_2 reads a split-planar tensor
    if (c->srnet[22].d_proj) {
        // conv2d_transpose_2 + ReLU + the 64 x 27 tap projection of conv2d_36 in one kernel ...
        ConvIO up2 = {b("up1"), 2 * H, 2 * W, b("up2"), 32, 32, 1, nullptr, 0, A_RELU, u2p, 0, c->srnet[22].d_proj};
        up2.in_split = u2p;
        if (c->fuse_gather && u2p && c->srnet[22].d_w3 && !g_conv_dbg) {
            // conv_tail.hip: the same three layers plane-major on a trunk-style kernel, then the border fix-up on its 16 x 64 tiles
            const ConvLayer& L = c->srnet[22];
            const size_t left = (size_t)(c->max_batch - c->lane_b0);
            const size_t tail_in = left * H * W * 1024;      // behind the lane's part of "up1": [left][4 planes][2H][2W][64 B]
            if (tail_in + 4096 > 4294967295.0) return fail(c, SR_ERR_ARG, "srnet.up2: tensor too large for 32-bit offsets");
            TailArgs ta = {(const unsigned char*)b("up1"), (const unsigned char*)L.d_w3, L.d_bias, (const unsigned char*)L.d_proj,
                           c->srnet[23].d_bias, d_out, (float*)b("up2"), (unsigned)tail_in, B, 2 * H, 2 * W, A_RELU, 3, 3, A_CLIP01, 0, 0};
            ConvArgs fx = {};
            fx.out = b("up2");
            fx.gather_out = d_out;
            fx.gather_bias = c->srnet[23].d_bias;
            fx.gather_cs = fx.gather_cout = 3;
            fx.gather_act = A_CLIP01;
            fx.B = B;
            fx.H = 2 * H;
            fx.W = 2 * W;
            const double px = (double)B * 4 * H * W;
            ProfScope ps(c, "srnet.up2", px * 9.0 * 64 * 64 * 2.0, px * 256.0 + 4 * px * (12.0 + 2.0 * 16.0 * (156 + 164) / 1024.0) + 2.0 * trunk_weight_bytes());
            LAUNCHCHK(c, "srnet.up2", launch_tail(ta, st));
            LAUNCHCHK(c, "srnet.up2.fix", launch_gather_fix(fx, st, 8));
            return SR_OK;
        }
        if (c->fuse_gather) {
            // ... and conv2d_36's nine-tap sum + clip as well: the kernel writes the image, except the tile-border pixels, whose
            // partial sums ("up2": 4 x 16 B per HR pixel, sparsely written) a small second kernel adds (ConvArgs::gather_out)
            up2.gather_out = d_out;
            up2.gather_bias = c->srnet[23].d_bias;
            up2.gather_cs = 3;
            up2.gather_cout = 3;
            up2.gather_act = A_CLIP01;
            return run_conv(c, P, st, c->srnet[22], B, up2, "srnet.up2");
        }
        // SR_NO_FUSE_GATHER=1: "up2" holds P[B,4H,4W,32] (128 B per HR pixel instead of 256 B of activations), then the 9-tap gather
        rc = run_conv(c, P, st, c->srnet[22], B, up2, "srnet.up2");
        if (rc) return rc;
        ConvArgs g = {};
        g.in = b("up2");
        g.bias = c->srnet[23].d_bias;
        g.out = d_out;
        g.B = B;
        g.H = 4 * H;
        g.W = 4 * W;
        g.out_cs = 3;
        g.cout_store = 3;
        g.out_f32 = 1;
        g.act = A_CLIP01;
        g.tiles_x = (g.W + TW - 1) / TW;
        g.tiles_y = (g.H + 7) / 8;
        const double px = (double)B * g.H * g.W;
        ProfScope ps(c, "srnet.out", px * 27 * 2, px * (128 + 12));
        LAUNCHCHK(c, "srnet.out", launch_tap_gather(g, st));
        return SR_OK;
    }
    ConvIO up2 = {b("up1"), 2 * H, 2 * W, b("up2"), 64, 64, af, nullptr, 0, A_RELU, u2p, 0};
    up2.in_split = u2p;
    rc = run_conv(c, P, st, c->srnet[22], B, up2, "srnet.up2");
    if (rc) return rc;
    return run_conv(c, P, st, c->srnet[23], B, {b("up2"), 4 * H, 4 * W, d_out, 3, 3, 1, nullptr, 0, A_CLIP01}, "srnet.out");
}

// one generator step on device buffers (reference FRVSR.py:58-63, TecoGAN.py:68-76)
int run_step_lane(sr_ctx* c, int B, const float* d_lr_t, const float* d_lr_tm1, const float* d_est_prev,
                  float* d_est_out, float* d_lr_warp, float* d_hr_warp, bool warp_only) {
    const int bf = c->prec == P_BF16;
    const int H = c->h, W = c->w;
    hipStream_t st = lane_st(c);
    float* d_flow = c->d_flow + (size_t)c->lane_b0 * H * W * 2;
    const bool split = c->step_split;   // decided per step (run_step), the same for every lane
    int rc = run_fnet(c, B, d_lr_t, d_lr_tm1, d_flow);
    if (rc) return rc;
    if (d_lr_warp) {
        ProfScope ps(c, "lr_warp", 0, (double)B * H * W * 4 * (3 + 2 + 3));
        LAUNCHCHK(c, "lr_warp", launch_warp3(d_lr_tm1, d_flow, d_lr_warp, B, H, W, st));
    }
    {
        // algorithmic bytes per LR pixel: read 2 flow + 48 prev-est + 3 LR (fp32), write 64 activations
        const double bytes = (double)B * H * W * (4.0 * (2 + 48 + 3) + 64.0 * (bf ? 2 : 4) + (d_hr_warp ? 192.0 : 0.0));
        ProfScope ps(c, "pre_srnet", 0, bytes);
        LAUNCHCHK(c, "pre_srnet", launch_pre_srnet(d_flow, d_est_prev, d_lr_t, lane_buf(c, "s_in"), d_hr_warp, split ? 2 : bf, B, H, W, st));
    }
    if (warp_only) return SR_OK;
    rc = run_srnet(c, B, d_est_out, split);
    if (rc) return rc;
    if (c->model == SR_TECOGAN) {
        ProfScope ps(c, "teco_epilogue", 0, (double)B * H * W * (12.0 + 2 * 192.0));
        LAUNCHCHK(c, "teco_epilogue", launch_teco_epilogue(d_est_out, d_lr_t, B, H, W, st));
    }
    return SR_OK;
}

int run_step(sr_ctx* c, int B, const float* d_lr_t, const float* d_lr_tm1, const float* d_est_prev,
             float* d_est_out, float* d_lr_warp, float* d_hr_warp, bool warp_only) {
    // profiling wants each kernel alone on the device; one image cannot be split
    // small steps (the 4 x 64 x 64 recurrence of a training iteration) are launch-bound: a second lane only
    // doubles the launches (measured 14.6 -> 12.9 ms per iteration with one lane)
    const bool tiny = (size_t)B * c->h * c->w <= 65536;
    const int lanes = tiny ? 1 : std::min(c->dual + 1, B);
    c->step_split = use_split(c, B);
    c->step_fnet_split = use_fnet_split(c, B);
    if (lanes < 2 || c->prof_on)
        return run_step_lane(c, B, d_lr_t, d_lr_tm1, d_est_prev, d_est_out, d_lr_warp, d_hr_warp, warp_only);
    const size_t lr = (size_t)c->h * c->w * 3, hr = lr * 16;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (int l = 1; l < lanes; ++l) HIPCHK(c, hipStreamWaitEvent(c->stream2[l - 1], c->ev_fork, 0));
    int rc = SR_OK, b0 = 0;
    for (int lane = 0; lane < lanes && rc == SR_OK; ++lane) {
        const int nb = B / lanes + (lane < B % lanes ? 1 : 0);
        c->lane_b0 = b0;
        c->lane_stream = lane == 0 ? c->stream : c->stream2[lane - 1];
        rc = run_step_lane(c, nb, d_lr_t + b0 * lr, d_lr_tm1 + b0 * lr, d_est_prev + b0 * hr,
                           d_est_out ? d_est_out + b0 * hr : nullptr, d_lr_warp ? d_lr_warp + b0 * lr : nullptr,
                           d_hr_warp ? d_hr_warp + b0 * hr : nullptr, warp_only);
        b0 += nb;
    }
    c->lane_b0 = 0;
    c->lane_stream = nullptr;
    // always join, also after a failed launch, so that `stream` stays ordered behind the other lanes
    bool ok = true;
    for (int l = 1; l < lanes; ++l) {
        ok &= hipEventRecord(c->ev_join[l - 1], c->stream2[l - 1]) == hipSuccess;
        ok &= hipStreamWaitEvent(c->stream, c->ev_join[l - 1], 0) == hipSuccess;
    }
    if (rc) return rc;
    if (!ok) return fail(c, SR_ERR_HIP, "stream join failed");
    return SR_OK;
}

// ---------------------------------------------------------------- caller tensors <-> kernel tensors
// The caller's tensors are dense (B, scale*h_in, scale*w_in, ch); the kernels' are (B, scale*h, scale*w, ch).  They
// are the same thing unless the ctx was made by sr_create_padded for a size that is not a multiple of 8: then
// inputs are zero padded at the bottom / right on the way in and outputs cropped on the way out (host tensors
// through a dense device staging buffer; stream order makes one buffer per size class enough).
int put_in(sr_ctx* c, float* d_dst, const float* src, bool src_dev, int B, int scale, int ch) {
    hipStream_t st = c->stream;
    const int rows = scale * c->h_in, cols = scale * c->w_in;
    const size_t n = (size_t)B * rows * cols * ch * 4;
    if (!c->pad) {
        HIPCHK(c, hipMemcpyAsync(d_dst, src, n, src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        return SR_OK;
    }
    const float* dsrc = src;
    if (!src_dev) {
        float* stage = (size_t)rows * cols * ch > (size_t)c->h_in * c->w_in * 3 ? c->d_io_hr : c->d_io_lr;
        HIPCHK(c, hipMemcpyAsync(stage, src, n, hipMemcpyHostToDevice, st));
        dsrc = stage;
    }
    LAUNCHCHK(c, "pad_copy", launch_pad_copy(dsrc, d_dst, B, rows, cols, ch, scale * c->h, scale * c->w, st));
    return SR_OK;
}

// submits the float read-back that sr_clip_read_est_begin left pending (see there)
int est_flush(sr_ctx* c) {
    if (!c->est_copy_dst) return SR_OK;
    HIPCHK(c, hipMemcpyAsync(c->est_copy_dst, c->est_copy_src, c->est_copy_bytes, hipMemcpyDeviceToHost, c->est_stream));
    HIPCHK(c, hipEventRecord(c->ev_est_done, c->est_stream));
    c->est_copy_dst = nullptr;
    return SR_OK;
}

int get_out(sr_ctx* c, float* dst, bool dst_dev, const float* d_src, int B, int scale, int ch) {
    hipStream_t st = c->stream;
    const int rows = scale * c->h_in, cols = scale * c->w_in;
    const size_t n = (size_t)B * rows * cols * ch * 4;
    if (!c->pad) {
        HIPCHK(c, hipMemcpyAsync(dst, d_src, n, dst_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
        return SR_OK;
    }
    float* stage = (size_t)rows * cols * ch > (size_t)c->h_in * c->w_in * 3 ? c->d_io_hr : c->d_io_lr;
    if (c->est_pending && !dst_dev && stage == c->d_io_hr) {   // sr_clip_read_est_begin is reading it
        int rc = est_flush(c);
        if (rc) return rc;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_est_done, 0));
    }
    float* ddst = dst_dev ? dst : stage;
    LAUNCHCHK(c, "crop_copy", launch_crop_copy(d_src, ddst, B, rows, cols, ch, scale * c->h, scale * c->w, st));
    if (!dst_dev) HIPCHK(c, hipMemcpyAsync(dst, stage, n, hipMemcpyDeviceToHost, st));
    return SR_OK;
}

// the uint8 output path's lazily made objects (sr_clip_push_u8); also the cleanup of a partially failed first use
void free_u8_path(sr_ctx* c) {
    for (int i = 0; i < 2; ++i) {
        if (c->d_u8[i]) (void)hipFree(c->d_u8[i]);
        if (c->ev_u8_ready[i]) (void)hipEventDestroy(c->ev_u8_ready[i]);
        if (c->ev_u8_done[i]) (void)hipEventDestroy(c->ev_u8_done[i]);
        c->d_u8[i] = nullptr;
        c->ev_u8_ready[i] = c->ev_u8_done[i] = nullptr;
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    c->copy_stream = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (c->d_in[i]) (void)hipFree(c->d_in[i]);
        if (c->ev_in_ready[i]) (void)hipEventDestroy(c->ev_in_ready[i]);
        if (c->ev_in_free[i]) (void)hipEventDestroy(c->ev_in_free[i]);
        c->d_in[i] = nullptr;
        c->ev_in_ready[i] = c->ev_in_free[i] = nullptr;
    }
    if (c->in_stream) (void)hipStreamDestroy(c->in_stream);
    c->in_stream = nullptr;
    c->u8_ready = false;
    c->u8_pushes = 0;
    c->u8_pending_host = nullptr;
    if (c->est_stream) (void)hipStreamDestroy(c->est_stream);
    if (c->ev_est_src) (void)hipEventDestroy(c->ev_est_src);
    if (c->ev_est_done) (void)hipEventDestroy(c->ev_est_done);
    c->est_stream = nullptr;
    c->ev_est_src = c->ev_est_done = nullptr;
    c->est_guard = 0;
    c->est_pending = false;
    c->est_copy_dst = nullptr;
}

}  // namespace srh

using namespace srh;

// =============================================================== C ABI
extern "C" {

const char* sr_last_error(const sr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int sr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_common(sr_ctx** out, int device, int h_in, int w_in, int ch, int scale, int max_batch, int dtype, int model,
                         bool allow_pad) {
    if (!out) return fail(nullptr, SR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (ch != 3) return fail(nullptr, SR_ERR_SHAPE, "channels must be 3, got %d", ch);
    if (scale != 4) return fail(nullptr, SR_ERR_SHAPE, "upscale must be 4, got %d", scale);
    if (!allow_pad && (h_in < 8 || w_in < 8 || h_in % 8 || w_in % 8))
        return fail(nullptr, SR_ERR_SHAPE, "LR tile %dx%d: height and width must be multiples of 8 "
                    "(FNet pools 3x and upsamples 3x; reference layers.py:31-41); sr_create_padded pads for you", h_in, w_in);
    if (h_in < 2 || w_in < 2) return fail(nullptr, SR_ERR_SHAPE, "LR tile %dx%d: dense_image_warp needs at least 2x2", h_in, w_in);
    const int h = round_up(h_in, 8), w = round_up(w_in, 8);
    if (max_batch < 1) return fail(nullptr, SR_ERR_ARG, "max_batch must be >= 1");
    if (dtype < 0 || dtype > 3) return fail(nullptr, SR_ERR_ARG, "dtype must be 0 (fp32), 1 (bf16), 2 (bf16x3) or 3 (f16mx6)");
    const bool mx6 = dtype == SR_F16MX6;
    if (mx6) dtype = SR_BF16X3;      // everything but the residual blocks' kernels
    if (model != SR_FRVSR && model != SR_TECOGAN) return fail(nullptr, SR_ERR_ARG, "model must be 0 (frvsr) or 1 (tecogan)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, SR_ERR_HIP, "no HIP device available (%s); libsr_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, SR_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    sr_ctx* c = new sr_ctx();
    c->device = device;
    c->h = h;
    c->w = w;
    c->h_in = h_in;
    c->w_in = w_in;
    c->pad = h != h_in || w != w_in;
    c->max_batch = max_batch;
    c->prec = dtype;
    c->model = model;
    if (const char* e = getenv("SR_TRUNK_SUB")) c->trunk_sub = atoi(e);
    if (getenv("SR_NO_PLANAR")) c->planar_trunk = 0;
    c->fuse_pool = getenv("SR_NO_FUSE_POOL") ? 0 : 1;
    c->fuse_gather = getenv("SR_NO_FUSE_GATHER") ? 0 : 1;
    c->trunk_up1 = getenv("SR_NO_TRUNK_UP1") ? 0 : 1;
    if (const char* e = getenv("SR_TAIL")) c->tail_kernel = atoi(e);
    if (const char* e = getenv("SR_SPLIT_TRUNK")) c->split_trunk = atoi(e);
    if (const char* e = getenv("SR_FNET")) c->fnet_split = atoi(e);
    c->trunk_mx6 = mx6 ? 1 : 0;
    if (const char* e = getenv("SR_MX6")) c->trunk_mx6 = (dtype == SR_BF16X3 && atoi(e)) ? 1 : 0;
    int rc = SR_OK;
    auto bail = [&](int code) {
        g_err = c->err;
        sr_destroy(c);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(c, SR_ERR_HIP, "hipSetDevice(%d) failed", device));
    if (hipStreamCreate(&c->stream) != hipSuccess) return bail(fail(c, SR_ERR_HIP, "hipStreamCreate failed"));
    c->own_stream = true;
    if (getenv("SR_NO_DUAL")) c->dual = 0;
    if (const char* e = getenv("SR_LANES")) c->dual = std::max(0, std::min(3, atoi(e) - 1));
    bool lanes_ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; l < c->dual; ++l)
        lanes_ok = lanes_ok && hipStreamCreateWithFlags(&c->stream2[l], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&c->ev_join[l], hipEventDisableTiming) == hipSuccess;
    if (!lanes_ok)
        return bail(fail(c, SR_ERR_HIP, "second stream / events could not be created"));
    const size_t es = dtype == SR_BF16 ? 2 : 4;
    const size_t B = max_batch, L0 = (size_t)h * w, L1 = L0 / 4, L2 = L0 / 16, L3 = L0 / 64;
    struct { const char* n; size_t elems; } plan[] = {
        {"x0", L0 * 16}, {"a0", L0 * 32}, {"b0", L0 * 32}, {"p1", L1 * 32}, {"a1", L1 * 64}, {"b1", L1 * 64},
        {"p2", L2 * 64}, {"a2", L2 * 128}, {"b2", L2 * 128}, {"p3", L3 * 128}, {"a3", L3 * 256}, {"b3", L3 * 256},
        {"u2", L2 * 256}, {"u1", L1 * 128}, {"u0", L0 * 64},
        {"s_in", L0 * 64}, {"sx", L0 * 64}, {"sy", L0 * 64}, {"up1", L0 * 4 * 64}, {"up2", L0 * 16 * 64}};
    for (auto& p : plan)
        if ((rc = alloc_buf(c, p.n, B * p.elems * es))) return bail(rc);
    struct { float** p; size_t elems; } fplan[] = {
        {&c->d_lr_t, L0 * 3}, {&c->d_lr_tm1, L0 * 3}, {&c->d_est[0], L0 * 48}, {&c->d_est[1], L0 * 48},
        {&c->d_flow, L0 * 2}, {&c->d_lr_warp, L0 * 3}, {&c->d_hr_warp, L0 * 48}, {&c->d_ss, L0 * 48}};
    for (auto& p : fplan) {
        if (hipMalloc((void**)p.p, B * p.elems * 4) != hipSuccess)
            return bail(fail(c, SR_ERR_HIP, "hipMalloc of %zu bytes failed", B * p.elems * 4));
    }
    if (c->pad) {   // caller-shaped staging for host <-> device copies
        const size_t n = B * (size_t)h_in * w_in * 48 * 4;
        if (hipMalloc((void**)&c->d_io_hr, n) != hipSuccess || hipMalloc((void**)&c->d_io_lr, n / 16) != hipSuccess)
            return bail(fail(c, SR_ERR_HIP, "hipMalloc of %zu bytes failed", n));
    }
    *out = c;
    return SR_OK;
}

int sr_create(sr_ctx** out, int device, int h, int w, int ch, int scale, int max_batch, int dtype, int model) {
    return create_common(out, device, h, w, ch, scale, max_batch, dtype, model, false);
}

int sr_create_padded(sr_ctx** out, int device, int h, int w, int ch, int scale, int max_batch, int dtype, int model) {
    return create_common(out, device, h, w, ch, scale, max_batch, dtype, model, true);
}

int sr_destroy(sr_ctx* c) {
    if (!c) return SR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int l = 0; l < 3; ++l) {
        if (c->stream2[l]) {
            (void)hipStreamSynchronize(c->stream2[l]);
            (void)hipStreamDestroy(c->stream2[l]);
        }
        if (c->ev_join[l]) (void)hipEventDestroy(c->ev_join[l]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    prof_resolve(c);
    train_free(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto& kv : c->buf) (void)hipFree(kv.second);
    float* fl[] = {c->d_lr_t, c->d_lr_tm1, c->d_est[0], c->d_est[1], c->d_flow, c->d_lr_warp, c->d_hr_warp, c->d_ss,
                   c->d_io_lr, c->d_io_hr};
    for (auto p : fl)
        if (p) (void)hipFree(p);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    free_u8_path(c);
    for (auto& L : c->fnet) free_layer(L);
    for (auto& L : c->srnet) free_layer(L);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SR_OK;
}

int sr_set_stream(sr_ctx* c, void* hip_stream) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->own_stream = false;
    c->stream = (hipStream_t)hip_stream;
    return SR_OK;
}

static int u8_flush(sr_ctx* c);

int sr_sync(sr_ctx* c) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = u8_flush(c);      // the newest uint8 frame's copy is still pending (sr_clip_push_u8)
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    return SR_OK;
}

int sr_load_tensor(sr_ctx* c, const char* name, const float* host, const int64_t* shape, int rank) {
    if (!c || !name || !host || !shape) return fail(c, SR_ERR_ARG, "NULL argument");
    if (rank != 1 && rank != 4) return fail(c, SR_ERR_WEIGHTS, "%s: rank %d, expected 1 (bias) or 4 (kernel)", name, rank);
    const bool is_f = strncmp(name, "FNet/", 5) == 0, is_s = strncmp(name, "SRNet/", 6) == 0;
    if (!is_f && !is_s) return fail(c, SR_ERR_WEIGHTS, "%s: tensor name must start with FNet/ or SRNet/", name);
    if (c->committed) {   // a new load_weights after a commit starts over
        c->tensors.clear();
        c->n_fnet = c->n_srnet = 0;
        c->committed = false;
    }
    if (is_f && c->n_srnet) return fail(c, SR_ERR_WEIGHTS, "%s: FNet tensors must precede SRNet tensors", name);
    int& n = is_f ? c->n_fnet : c->n_srnet;
    const int lim = is_f ? 28 : 48;
    if (n >= lim) return fail(c, SR_ERR_WEIGHTS, "%s: more than %d tensors in group", name, lim);
    const int layer = n / 2;
    const bool is_kernel = (n % 2) == 0;
    int cin, cout, mode = M_CONV;
    if (is_f) {
        cin = FNET_SPEC[layer][0];
        cout = FNET_SPEC[layer][1];
    } else {
        srnet_spec(layer, cin, cout, mode);
    }
    if (is_kernel) {
        const int64_t want[4] = {3, 3, mode == M_CONV ? cin : cout, mode == M_CONV ? cout : cin};
        if (rank != 4 || memcmp(shape, want, sizeof want))
            return fail(c, SR_ERR_WEIGHTS, "%s: position %d of its group expects a kernel of shape (3,3,%lld,%lld)",
                        name, n, (long long)want[2], (long long)want[3]);
    } else if (rank != 1 || shape[0] != cout) {
        return fail(c, SR_ERR_WEIGHTS, "%s: position %d of its group expects a bias of shape (%d,)", name, n, cout);
    }
    HostTensor t;
    t.name = name;
    t.shape.assign(shape, shape + rank);
    size_t cnt = 1;
    for (int i = 0; i < rank; ++i) cnt *= (size_t)shape[i];
    t.data.assign(host, host + cnt);
    c->tensors.push_back(std::move(t));
    ++n;
    return SR_OK;
}

int sr_commit_weights(sr_ctx* c) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (c->n_fnet != 28 || c->n_srnet != 48)
        return fail(c, SR_ERR_WEIGHTS, "expected 28 FNet + 48 SRNet tensors, have %d + %d", c->n_fnet, c->n_srnet);
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 14; ++i) {
        const int lvl[14] = {0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 1, 1, 0, 0};
        plan_layer(c->fnet[i], c->prec, M_CONV, FNET_SPEC[i][0], FNET_SPEC[i][1], conv_tiles(c->max_batch, c->h >> lvl[i], c->w >> lvl[i]));
        int rc = upload_layer(c, c->fnet[i], c->prec, c->tensors[2 * i].data.data(), c->tensors[2 * i + 1].data.data());
        if (rc) return rc;
    }
    for (int i = 0; i < 24; ++i) {
        int cin, cout, mode;
        srnet_spec(i, cin, cout, mode);
        plan_layer(c->srnet[i], c->prec, mode, cin, cout, conv_tiles(c->max_batch, c->h << (i == 23 ? 2 : 0), c->w << (i == 23 ? 2 : 0)));
        int rc = upload_layer(c, c->srnet[i], c->prec, c->tensors[28 + 2 * i].data.data(), c->tensors[28 + 2 * i + 1].data.data());
        if (rc) return rc;
    }
    if (c->prec == P_BF16X3 && c->split_trunk) {   // split-planar trunk: second packing of the 21 trunk convs and of conv2d_transpose_1
        // conv2d_transpose_1: conv_trunk.hip's transposed-conv packing, or (SR_NO_TRUNK_UP1) the generic kernel's with the split
        // channel order; conv2d_transpose_2: the generic kernel's packing in the split channel order (its input is split-planar
        // when conv2d_transpose_1 runs on conv_trunk.hip)
        for (int i = 0; i <= 22; ++i) {
            ConvLayer& L = c->srnet[i];
            const bool tk = i < 21 || (i == 21 && c->trunk_up1);
            L.ents2 = i < 21 ? trunk_entries(L.cin, L.cout) : (tk ? trunkT_entries(L.cin, L.cout) : pack_entries(L, c->prec, PK_FWD, true));
            const size_t bytes = tk ? trunk_weight_bytes() : (size_t)L.ncb * L.wstride_cb * 16;
            int rc = upload_alt(c, L, bytes, c->tensors[28 + 2 * i].data.data());
            if (rc) return rc;
        }
    }
    c->mx6_valid = false;
    if (c->prec == P_BF16X3 && c->split_trunk && c->trunk_mx6) {   // conv_trunk6.hip: f16 + FP6 packing of the residual blocks' convs
        std::vector<unsigned char> packed(trunk6_weight_bytes());
        for (int i = 1; i <= 20; ++i) {
            ConvLayer& L = c->srnet[i];
            trunk6_pack_weights(c->tensors[28 + 2 * i].data.data(), packed.data());
            if (!L.d_w6) HIPCHK(c, hipMalloc(&L.d_w6, packed.size()));
            HIPCHK(c, hipMemcpy(L.d_w6, packed.data(), packed.size(), hipMemcpyHostToDevice));
        }
        c->mx6_valid = true;
    }
    if (c->prec == P_BF16X3 && c->fnet_split) {   // conv_fnet.hip: second packing of FNet's 13 MFMA convs (conv2d_14 stays on the taps-in-N kernel)
        for (int i = 0; i < 13; ++i) {
            ConvLayer& L = c->fnet[i];
            L.ents2 = fconv_entries(L.cin, L.cout);
            int rc = upload_alt(c, L, fnet_weight_bytes(L.cin, L.cout), c->tensors[2 * i].data.data());
            if (rc) return rc;
        }
    }
    if (c->prec == P_BF16X3 && c->split_trunk && c->trunk_up1 && c->tail_kernel) {   // conv_tail.hip reads conv2d_transpose_2 in trunkT order
        ConvLayer& L = c->srnet[22];
        L.ents3 = trunkT_entries(L.cin, L.cout);
        std::vector<uint8_t> packed = pack_apply(L.ents3, trunk_weight_bytes(), P_BF16X3, c->tensors[28 + 2 * 22].data.data());
        if (L.d_w3) (void)hipFree(L.d_w3);
        L.d_w3 = nullptr;
        HIPCHK(c, hipMalloc(&L.d_w3, packed.size()));
        HIPCHK(c, hipMemcpy(L.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
    }
    if (!getenv("SR_NO_FUSE_OUT")) {   // SRNet tail: conv2d_transpose_2 carries conv2d_36's tap projection
        int rc = upload_proj(c, c->srnet[22], c->prec, c->tensors[28 + 2 * 23].data.data(), 3);
        if (rc) return rc;
    }
    c->committed = true;
    return train_reload(c);
}

int sr_get_tensor(const sr_ctx* c, int index, char* name_out, int name_cap, float* host_out, int64_t* shape_out, int* rank_out) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (index < 0 || index >= (int)c->tensors.size()) return fail(nullptr, SR_ERR_ARG, "tensor index %d out of range", index);
    if (int rc = train_sync_host(const_cast<sr_ctx*>(c))) return rc;   // weights updated by sr_train_apply live on the device
    const HostTensor& t = c->tensors[index];
    if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", t.name.c_str());
    if (rank_out) *rank_out = (int)t.shape.size();
    if (shape_out) memcpy(shape_out, t.shape.data(), t.shape.size() * sizeof(int64_t));
    if (host_out) memcpy(host_out, t.data.data(), t.data.size() * 4);
    return SR_OK;
}

int sr_step_dev(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, float* est_t, float* lr_warp) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (!lr_t || !lr_tm1 || !est_tm1 || !est_t) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    if (!c->pad) return run_step(c, B, lr_t, lr_tm1, est_tm1, est_t, lr_warp, nullptr, false);
    c->clip_active = false;   // pad mode borrows the state buffers as the padded tensors
    if ((rc = put_in(c, c->d_lr_t, lr_t, true, B, 1, 3)) || (rc = put_in(c, c->d_lr_tm1, lr_tm1, true, B, 1, 3)) ||
        (rc = put_in(c, c->d_est[0], est_tm1, true, B, 4, 3)))
        return rc;
    rc = run_step(c, B, c->d_lr_t, c->d_lr_tm1, c->d_est[0], c->d_est[1], lr_warp ? c->d_lr_warp : nullptr, nullptr, false);
    if (rc) return rc;
    if ((rc = get_out(c, est_t, true, c->d_est[1], B, 4, 3))) return rc;
    return lr_warp ? get_out(c, lr_warp, true, c->d_lr_warp, B, 1, 3) : SR_OK;
}

int sr_step(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, float* est_t, float* lr_warp) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (!lr_t || !lr_tm1 || !est_tm1 || !est_t) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    c->clip_active = false;   // the stateless step borrows the state buffers
    hipStream_t st = c->stream;
    if ((rc = put_in(c, c->d_lr_t, lr_t, false, B, 1, 3)) || (rc = put_in(c, c->d_lr_tm1, lr_tm1, false, B, 1, 3)) ||
        (rc = put_in(c, c->d_est[0], est_tm1, false, B, 4, 3)))
        return rc;
    rc = run_step(c, B, c->d_lr_t, c->d_lr_tm1, c->d_est[0], c->d_est[1], lr_warp ? c->d_lr_warp : nullptr, nullptr, false);
    if (rc) return rc;
    if ((rc = get_out(c, est_t, false, c->d_est[1], B, 4, 3))) return rc;
    if (lr_warp && (rc = get_out(c, lr_warp, false, c->d_lr_warp, B, 1, 3))) return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    return SR_OK;
}

int sr_step_hr_warp(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, const float* est_tm1, float* hr_warp) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (!lr_t || !lr_tm1 || !est_tm1 || !hr_warp) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    c->clip_active = false;
    hipStream_t st = c->stream;
    if ((rc = put_in(c, c->d_lr_t, lr_t, false, B, 1, 3)) || (rc = put_in(c, c->d_lr_tm1, lr_tm1, false, B, 1, 3)) ||
        (rc = put_in(c, c->d_est[0], est_tm1, false, B, 4, 3)))
        return rc;
    rc = run_step(c, B, c->d_lr_t, c->d_lr_tm1, c->d_est[0], nullptr, nullptr, c->d_hr_warp, true);
    if (rc) return rc;
    if ((rc = get_out(c, hr_warp, false, c->d_hr_warp, B, 4, 3))) return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    return SR_OK;
}

int sr_clip_begin(sr_ctx* c, int B) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    const size_t nlr = (size_t)B * c->h * c->w * 3 * 4;
    if (c->est_pending) {      // a float read-back of the previous clip's last estimates that has not been ended: it reads what the
        if ((rc = est_flush(c))) return rc;                                  // memsets below clear
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_est_done, 0));
    }
    // state0 = zeros for prev LR and prev estimate (reference FRVSR.py:233-234)
    HIPCHK(c, hipMemsetAsync(c->d_lr_tm1, 0, nlr, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_est[0], 0, nlr * 16, c->stream));
    if ((rc = u8_flush(c))) return rc;      // a previous clip's last uint8 frame, if its caller never waited for it
    c->cur = 0;
    c->clip_B = B;
    c->clip_active = true;
    return SR_OK;
}

static int clip_push_common(sr_ctx* c, const float* lr_t, float* est_t, bool dev, bool sync) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->clip_active) return fail(c, SR_ERR_STATE, "sr_clip_push without sr_clip_begin");
    if (!lr_t) return fail(c, SR_ERR_ARG, "lr_t is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const int B = c->clip_B;
    hipStream_t st = c->stream;
    // a float read-back left pending by sr_clip_read_est_begin: submitted now, behind this push's own input copy if it has one
    int rc = est_flush(c);
    if (rc) return rc;
    // sr_clip_read_est_begin is still copying the estimate of two pushes ago out of the buffer this step writes
    if (c->est_guard > 0 && --c->est_guard == 0) HIPCHK(c, hipStreamWaitEvent(st, c->ev_est_done, 0));
    rc = put_in(c, c->d_lr_t, lr_t, dev, B, 1, 3);
    if (rc) return rc;
    float* out = c->d_est[c->cur ^ 1];
    rc = run_step(c, B, c->d_lr_t, c->d_lr_tm1, c->d_est[c->cur], out, nullptr, nullptr, false);
    if (rc) return rc;
    // est state := output, prev LR := current (reference FRVSR.py:293,309)
    c->cur ^= 1;
    float* t = c->d_lr_t;
    c->d_lr_t = c->d_lr_tm1;
    c->d_lr_tm1 = t;
    if (est_t && (rc = get_out(c, est_t, dev, out, B, 4, 3))) return rc;
    if (sync) HIPCHK(c, hipStreamSynchronize(st));
    return SR_OK;
}

int sr_clip_push(sr_ctx* c, const float* lr_t, float* est_t) { return clip_push_common(c, lr_t, est_t, false, true); }
int sr_clip_push_dev(sr_ctx* c, const float* lr_t, float* est_t) { return clip_push_common(c, lr_t, est_t, true, false); }

// submits the pending uint8 frame's copy to the host (the kernel cropped rows and columns: one dense copy)
static int u8_flush(sr_ctx* c) {
    if (!c->u8_pending_host) return SR_OK;
    const int slot = c->u8_pending_slot;
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_u8_ready[slot], 0));
    HIPCHK(c, hipMemcpyAsync(c->u8_pending_host, c->d_u8[slot], c->u8_pending_bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_u8_done[slot], c->copy_stream));
    c->u8_pending_host = nullptr;
    return SR_OK;
}

int sr_clip_push_u8(sr_ctx* c, const float* lr_t, int lr_on_device, uint8_t* est_u8_host) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!est_u8_host) return fail(c, SR_ERR_ARG, "est_u8_host is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->clip_active) return fail(c, SR_ERR_STATE, "sr_clip_push_u8 without sr_clip_begin");   // before anything is allocated
    if (!c->u8_ready) {   // first use: the copy stream, two device frames and their events; ready only when ALL of them exist
        const size_t n = (size_t)c->max_batch * 16 * c->h_in * c->w_in * 3;
        bool ok = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&c->in_stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipMalloc((void**)&c->d_u8[i], n) == hipSuccess &&
                 hipMalloc((void**)&c->d_in[i], (size_t)c->max_batch * c->h_in * c->w_in * 3 * sizeof(float)) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_u8_ready[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_u8_done[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_in_ready[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_in_free[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {   // leave nothing half made: the next call starts over
            (void)hipGetLastError();
            free_u8_path(c);
            return fail(c, SR_ERR_HIP, "uint8 output path: stream / buffer / event creation failed");
        }
        c->u8_ready = true;
    }
    const int B = c->clip_B, slot = (int)(c->u8_pushes & 1);
    if (!lr_t) return fail(c, SR_ERR_ARG, "lr_t is NULL");
    // Host frames go to the device on a stream of their own, now, under the previous push's kernels, into d_in[slot] (free
    // again when the step that used it two pushes ago has copied it on); on the ctx stream the copy would wait for those
    // kernels, 0.2 ms per step in the open.  The copy engine serves copies in SUBMISSION order, so the previous frame's
    // uint8 D2H -- which waits for that frame's kernels -- must be submitted AFTER this copy, not before: it is kept pending by
    // the push that produced it and submitted here (or by sr_u8_wait(ctx, 0)).  Submitted first, it held this copy and with it
    // the whole step back by a D2H (measured 7.1 instead of 6.0 ms per step); a copy kernel reading the page-locked frames
    // instead starved behind the trunk kernels, which own every register of the CUs they run on (7.2 ms).
    bool early = false;
    if (!lr_on_device) {
        if (c->u8_pushes >= 2) HIPCHK(c, hipStreamWaitEvent(c->in_stream, c->ev_in_free[slot], 0));
        HIPCHK(c, hipMemcpyAsync(c->d_in[slot], lr_t, (size_t)B * c->h_in * c->w_in * 3 * sizeof(float), hipMemcpyHostToDevice, c->in_stream));
        HIPCHK(c, hipEventRecord(c->ev_in_ready[slot], c->in_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in_ready[slot], 0));
        lr_t = c->d_in[slot];
        early = true;
    }
    int rc0 = u8_flush(c);
    if (rc0) return rc0;
    int rc = clip_push_common(c, lr_t, nullptr, lr_on_device != 0 || early, false);   // no fp32 copy out, no synchronisation
    if (early) HIPCHK(c, hipEventRecord(c->ev_in_free[slot], c->stream));   // (also after a failed step: the slot's next user waits for it)
    if (rc) return rc;
    const float* est = c->d_est[c->cur];   // the estimate this push produced (state already swapped)
    hipStream_t st = c->stream;
    if (c->u8_pushes >= 2) HIPCHK(c, hipStreamWaitEvent(st, c->ev_u8_done[slot], 0));   // frame n-2 has left this slot
    LAUNCHCHK(c, "denorm_u8", launch_denorm_u8(est, c->d_u8[slot], B, 4 * c->h, 4 * c->w * 3, 4 * c->h_in, 4 * c->w_in * 3, st));
    HIPCHK(c, hipEventRecord(c->ev_u8_ready[slot], st));
    // its copy to the host: submitted by the next push behind that push's own input copy, or by sr_u8_wait(ctx, 0)
    c->u8_pending_host = est_u8_host;
    c->u8_pending_slot = slot;
    c->u8_pending_bytes = (size_t)B * 16 * c->h_in * c->w_in * 3;
    ++c->u8_pushes;
    return SR_OK;
}

int sr_u8_wait(sr_ctx* c, int keep_in_flight) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->copy_stream || c->u8_pushes == 0) return SR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = c->u8_pushes;
    if (keep_in_flight >= 1) {   // everything but the newest frame (whose copy the next push submits)
        if (n >= 2) HIPCHK(c, hipEventSynchronize(c->ev_u8_done[(n - 2) & 1]));
        return SR_OK;
    }
    int rc = u8_flush(c);
    if (rc) return rc;
    HIPCHK(c, hipEventSynchronize(c->ev_u8_done[(n - 1) & 1]));
    if (n >= 2) HIPCHK(c, hipEventSynchronize(c->ev_u8_done[(n - 2) & 1]));
    return SR_OK;
}

int sr_clip_read_est(sr_ctx* c, int first, int n, float* est) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->clip_active) return fail(c, SR_ERR_STATE, "sr_clip_read_est without sr_clip_begin");
    if (!est) return fail(c, SR_ERR_ARG, "est is NULL");
    if (first < 0 || n < 1 || first + n > c->clip_B) return fail(c, SR_ERR_ARG, "clips [%d, %d) outside the clip batch of %d", first, first + n, c->clip_B);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t img = (size_t)c->h * c->w * 48;   // kernel-side image (padded ctx: get_out crops)
    int rc = get_out(c, est, false, c->d_est[c->cur] + (size_t)first * img, n, 4, 3);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SR_OK;
}

int sr_clip_read_est_begin(sr_ctx* c, int first, int n, float* est) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->clip_active) return fail(c, SR_ERR_STATE, "sr_clip_read_est_begin without sr_clip_begin");
    if (!est) return fail(c, SR_ERR_ARG, "est is NULL");
    if (first < 0 || n < 1 || first + n > c->clip_B) return fail(c, SR_ERR_ARG, "clips [%d, %d) outside the clip batch of %d", first, first + n, c->clip_B);
    if (c->est_pending) return fail(c, SR_ERR_STATE, "sr_clip_read_est_begin: the previous read-back has not been ended");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->est_stream) {
        bool ok = hipStreamCreateWithFlags(&c->est_stream, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&c->ev_est_src, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&c->ev_est_done, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (c->est_stream) (void)hipStreamDestroy(c->est_stream);
            if (c->ev_est_src) (void)hipEventDestroy(c->ev_est_src);
            if (c->ev_est_done) (void)hipEventDestroy(c->ev_est_done);
            c->est_stream = nullptr;
            c->ev_est_src = c->ev_est_done = nullptr;
            return fail(c, SR_ERR_HIP, "estimate read-back: stream / event creation failed");
        }
    }
    const size_t img = (size_t)c->h * c->w * 48;
    const float* src = c->d_est[c->cur] + (size_t)first * img;
    const size_t bytes = (size_t)n * 16 * c->h_in * c->w_in * 3 * 4;
    if (c->pad) {   // crop on the step's stream into the dense staging tensor, copy that
        LAUNCHCHK(c, "crop_copy", launch_crop_copy(src, c->d_io_hr, n, 4 * c->h_in, 4 * c->w_in, 3, 4 * c->h, 4 * c->w, c->stream));
        src = c->d_io_hr;
    }
    HIPCHK(c, hipEventRecord(c->ev_est_src, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->est_stream, c->ev_est_src, 0));
    // The copy itself is submitted by the next push, behind that push's input copy (or by sr_clip_read_est_end): the copy
    // engine serves copies in submission order, and 177 MB of estimates in front of the next frames' 11 MB held that step back
    // by the whole read-back (3.5 ms on every metric frame of main.py's eval loop).
    c->est_copy_dst = est;
    c->est_copy_src = src;
    c->est_copy_bytes = bytes;
    // the next push reads this estimate, the one after it writes over it; a padded ctx copies from the staging tensor, which
    // no step touches (get_out waits for the copy before it re-uses that tensor)
    c->est_guard = c->pad ? 0 : 2;
    c->est_pending = true;
    return SR_OK;
}

int sr_clip_read_est_end(sr_ctx* c) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (!c->est_pending) return fail(c, SR_ERR_STATE, "sr_clip_read_est_end without sr_clip_read_est_begin");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = est_flush(c);
    if (rc) return rc;
    HIPCHK(c, hipEventSynchronize(c->ev_est_done));
    c->est_pending = false;
    return SR_OK;
}

int sr_host_alloc(void** out, size_t bytes) {
    if (!out) return fail(nullptr, SR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (bytes == 0) return fail(nullptr, SR_ERR_ARG, "sr_host_alloc of 0 bytes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SR_ERR_HIP, "no HIP device available; libsr_hip has no CPU fallback");
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(nullptr, SR_ERR_HIP, "hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    return SR_OK;
}

int sr_host_free(void* p) {
    if (!p) return SR_OK;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(nullptr, SR_ERR_HIP, "hipHostFree failed: %s", hipGetErrorString(e));
    return SR_OK;
}

int sr_fnet(sr_ctx* c, int B, const float* lr_t, const float* lr_tm1, float* flow) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (!lr_t || !lr_tm1 || !flow) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    c->clip_active = false;
    hipStream_t st = c->stream;
    if ((rc = put_in(c, c->d_lr_t, lr_t, false, B, 1, 3)) || (rc = put_in(c, c->d_lr_tm1, lr_tm1, false, B, 1, 3))) return rc;
    c->step_fnet_split = use_fnet_split(c, B);
    rc = run_fnet(c, B, c->d_lr_t, c->d_lr_tm1, c->d_flow);
    if (rc) return rc;
    if ((rc = get_out(c, flow, false, c->d_flow, B, 1, 2))) return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    return SR_OK;
}

int sr_srnet(sr_ctx* c, int B, const float* lr_t, const float* s_s, float* out) {
    int rc = check_ready(c, B);
    if (rc) return rc;
    if (!lr_t || !s_s || !out) return fail(c, SR_ERR_ARG, "NULL tensor pointer");
    c->clip_active = false;
    hipStream_t st = c->stream;
    if ((rc = put_in(c, c->d_lr_t, lr_t, false, B, 1, 3)) || (rc = put_in(c, c->d_ss, s_s, false, B, 1, 48))) return rc;
    LAUNCHCHK(c, "srnet.input", launch_srnet_input(c->d_lr_t, c->d_ss, c->buf["s_in"], c->prec == P_BF16, B, c->h, c->w, st));
    rc = run_srnet(c, B, c->d_est[1], false);   // srnet_input writes the NHWC trunk input
    if (rc) return rc;
    if ((rc = get_out(c, out, false, c->d_est[1], B, 4, 3))) return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    return SR_OK;
}

// ---------------------------------------------------------------- profiling
int sr_profile_enable(sr_ctx* c, int enable) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    c->prof_on = enable != 0;
    return SR_OK;
}
int sr_profile_count(const sr_ctx* c) { return c ? (int)c->prof.size() : 0; }
int sr_profile_read(sr_ctx* c, int i, char* name_out, int name_cap, double* ms_total, int64_t* launches,
                    double* flops_per_launch, double* bytes_per_launch) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    if (i < 0 || i >= (int)c->prof.size()) return fail(c, SR_ERR_ARG, "profile index out of range");
    (void)hipSetDevice(c->device);
    prof_resolve(c);
    const ProfEntry& p = c->prof[i];
    if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", p.name.c_str());
    if (ms_total) *ms_total = p.ms;
    if (launches) *launches = p.launches;
    const double n = p.launches > 0 ? (double)p.launches : 1.0;
    if (flops_per_launch) *flops_per_launch = p.flops / n;
    if (bytes_per_launch) *bytes_per_launch = p.bytes / n;
    return SR_OK;
}
int sr_profile_reset(sr_ctx* c) {
    if (!c) return fail(nullptr, SR_ERR_ARG, "ctx is NULL");
    (void)hipSetDevice(c->device);
    prof_resolve(c);
    c->prof.clear();
    return SR_OK;
}

// ---------------------------------------------------------------- single ops
namespace {
struct DevBufs {
    std::vector<void*> p;
    ~DevBufs() {
        for (auto q : p) (void)hipFree(q);
    }
    void* get(size_t bytes) {
        void* q = nullptr;
        if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
        p.push_back(q);
        return q;
    }
};

int op_prologue(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SR_ERR_HIP, "no HIP device available; libsr_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, SR_ERR_ARG, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SR_ERR_HIP, "hipSetDevice failed");
    return SR_OK;
}

int conv_op(int device, int dtype, int mode, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
            const float* bias, int act, const float* res, float* out) {
    int rc = op_prologue(device);
    if (rc) return rc;
    if (!x || !w || !bias || !out) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    if (dtype < 0 || dtype > 2) return fail(nullptr, SR_ERR_ARG, "bad dtype");
    if (act < 0 || act > 4) return fail(nullptr, SR_ERR_ARG, "bad activation");
    ConvLayer L;
    plan_layer(L, dtype, mode, Cin, Cout);
    struct Guard { ConvLayer& l; ~Guard() { free_layer(l); } } guard{L};
    if ((rc = upload_layer(nullptr, L, dtype, w, bias))) return rc;
    DevBufs d;
    const size_t npix = (size_t)B * H * W, opix = mode == M_CONV ? npix : 4 * npix;
    const size_t es = dtype == SR_BF16 ? 2 : 4;
    float* d_x = (float*)d.get(npix * Cin * 4);
    void* d_xp = d.get(npix * L.cin_p * es);
    float* d_out = (float*)d.get(opix * Cout * 4);
    float* d_res = res ? (float*)d.get(opix * Cout * 4) : nullptr;
    if (!d_x || !d_xp || !d_out || (res && !d_res)) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
    HIPCHK(nullptr, hipMemcpy(d_x, x, npix * Cin * 4, hipMemcpyHostToDevice));
    if (res) HIPCHK(nullptr, hipMemcpy(d_res, res, opix * Cout * 4, hipMemcpyHostToDevice));
    LAUNCHCHK(nullptr, "pad_channels", launch_pad_channels(d_x, d_xp, dtype == SR_BF16, (long)npix, Cin, L.cin_p, nullptr));
    rc = run_conv(nullptr, dtype, nullptr, L, B, {d_xp, H, W, d_out, Cout, Cout, 1, d_res, Cout, act}, "op_conv");
    if (rc) return rc;
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(out, d_out, opix * Cout * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

typedef hipError_t (*simple_launch)(const float*, float*, int, int, int, int, hipStream_t);

int simple_op(int device, size_t in_elems, size_t out_elems, const float* x, float* out,
              const std::function<hipError_t(const float*, float*)>& fn) {
    int rc = op_prologue(device);
    if (rc) return rc;
    if (!x || !out) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    DevBufs d;
    float* d_x = (float*)d.get(in_elems * 4);
    float* d_o = (float*)d.get(out_elems * 4);
    if (!d_x || !d_o) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
    HIPCHK(nullptr, hipMemcpy(d_x, x, in_elems * 4, hipMemcpyHostToDevice));
    LAUNCHCHK(nullptr, "op", fn(d_x, d_o));
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(out, d_o, out_elems * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}
}  // namespace

int sr_op_conv3x3(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                  const float* bias, int act, const float* res, float* out) {
    return conv_op(device, dtype, M_CONV, B, H, W, Cin, Cout, x, w, bias, act, res, out);
}

int sr_op_convT3x3s2(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* x, const float* w,
                     const float* bias, int act, float* out) {
    return conv_op(device, dtype, M_CONVT, B, H, W, Cin, Cout, x, w, bias, act, nullptr, out);
}

int sr_op_conv3x3_mx6(int device, int B, int H, int W, const float* x, const float* w, const float* bias, int relu, const float* res,
                      int mode, float* out) {
    int rc = op_prologue(device);
    if (rc) return rc;
    if (!x || !out) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 1 || W < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    if (mode < 0 || mode > 3) return fail(nullptr, SR_ERR_ARG, "mode must be 0..3");
    const size_t npix = (size_t)B * H * W, bytes = npix * 256;
    if ((double)bytes + 4096.0 > 4294967295.0) return fail(nullptr, SR_ERR_ARG, "tensor too large for 32-bit offsets");
    DevBufs d;
    float* d_f = (float*)d.get(npix * 64 * 4);
    unsigned char* d_in = (unsigned char*)d.get(bytes + 4096);
    unsigned char* d_o = (unsigned char*)d.get(bytes + 4096);
    unsigned char* d_r = (unsigned char*)d.get(bytes + 4096);
    unsigned char* d_w = (unsigned char*)d.get(trunk6_weight_bytes());
    float* d_b = (float*)d.get(64 * 4);
    if (!d_f || !d_in || !d_o || !d_r || !d_w || !d_b) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
    HIPCHK(nullptr, hipMemset(d_in + bytes, 0, 4096));
    HIPCHK(nullptr, hipMemset(d_o + bytes, 0, 4096));
    HIPCHK(nullptr, hipMemset(d_r + bytes, 0, 4096));
    HIPCHK(nullptr, hipMemcpy(d_f, x, npix * 64 * 4, hipMemcpyHostToDevice));
    LAUNCHCHK(nullptr, "nhwc_to_mx6", launch_nhwc_to_mx6(d_f, d_in, B, H, W, nullptr));
    if (mode >= 2) {     // the format alone: 2 = the stored value (hi + lo6), 3 = the x6 plane
        LAUNCHCHK(nullptr, "mx6_to_nhwc", launch_mx6_to_nhwc(d_in, d_f, B, H, W, mode - 2, nullptr));
        HIPCHK(nullptr, hipDeviceSynchronize());
        HIPCHK(nullptr, hipMemcpy(out, d_f, npix * 64 * 4, hipMemcpyDeviceToHost));
        return SR_OK;
    }
    if (!w || !bias) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if ((relu != 0) == (res != nullptr)) return fail(nullptr, SR_ERR_ARG, "a ReLU layer (relu, no res) or a residual layer (res, no relu)");
    if (mode == 1 && !res) return fail(nullptr, SR_ERR_ARG, "split-planar output is the residual layer's");
    if (res) {
        HIPCHK(nullptr, hipMemcpy(d_f, res, npix * 64 * 4, hipMemcpyHostToDevice));
        LAUNCHCHK(nullptr, "nhwc_to_mx6", launch_nhwc_to_mx6(d_f, d_r, B, H, W, nullptr));
    }
    std::vector<unsigned char> packed(trunk6_weight_bytes());
    trunk6_pack_weights(w, packed.data());
    HIPCHK(nullptr, hipMemcpy(d_w, packed.data(), packed.size(), hipMemcpyHostToDevice));
    HIPCHK(nullptr, hipMemcpy(d_b, bias, 64 * 4, hipMemcpyHostToDevice));
    // a residual layer of the trunk updates x in place
    TrunkArgs ta = {d_in, (res && mode == 0) ? d_r : d_o, res ? d_r : nullptr, d_w, d_b, (unsigned)bytes, 0u, B, H, W, relu, 0, 0, 0};
    LAUNCHCHK(nullptr, "trunk6", launch_trunk6_conv(ta, mode == 1, nullptr));
    if (mode == 1) {     // the split-planar records conv2d_transpose_1 reads
        LAUNCHCHK(nullptr, "split_to_nhwc", launch_split_to_nhwc(d_o, d_f, B, H, W, nullptr));
    } else {
        LAUNCHCHK(nullptr, "mx6_to_nhwc", launch_mx6_to_nhwc(ta.out, d_f, B, H, W, 0, nullptr));
    }
    HIPCHK(nullptr, hipDeviceSynchronize());
    HIPCHK(nullptr, hipMemcpy(out, d_f, npix * 64 * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_op_maxpool2(int device, int B, int H, int W, int C, const float* x, float* out) {
    if (B < 1 || H < 1 || W < 1 || C < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    const size_t n_in = (size_t)B * H * W * C, n_out = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    return simple_op(device, n_in, n_out, x, out, [&](const float* a, float* o) {
        // channel counts the hot path uses go through the vectorised kernel
        return C % 4 == 0 ? launch_maxpool2(a, o, 0, B, H, W, C, nullptr) : launch_maxpool2_generic(a, o, B, H, W, C, nullptr);
    });
}

int sr_op_resize_bilinear(int device, int B, int H, int W, int C, int s, const float* x, float* out) {
    if (B < 1 || H < 1 || W < 1 || C < 1 || s < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    const size_t n_in = (size_t)B * H * W * C;
    return simple_op(device, n_in, n_in * s * s, x, out, [&](const float* a, float* o) {
        return (s == 2 && C % 4 == 0) ? launch_resize2x(a, o, 0, B, H, W, C, nullptr)
                                      : launch_resize_generic(a, o, B, H, W, C, s, nullptr);
    });
}

int sr_op_warp(int device, int B, int H, int W, int C, const float* img, const float* flow, float* out) {
    if (B < 1 || H < 2 || W < 2 || C < 1) return fail(nullptr, SR_ERR_ARG, "dense_image_warp needs H, W >= 2");
    if (!flow) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    int rc = op_prologue(device);
    if (rc) return rc;
    DevBufs d;
    const size_t npix = (size_t)B * H * W;
    float* d_f = (float*)d.get(npix * 2 * 4);
    if (!d_f) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
    HIPCHK(nullptr, hipMemcpy(d_f, flow, npix * 2 * 4, hipMemcpyHostToDevice));
    return simple_op(device, npix * C, npix * C, img, out, [&](const float* a, float* o) {
        return launch_warp_generic(a, d_f, o, B, H, W, C, nullptr);
    });
}

int sr_op_space_to_depth(int device, int B, int H, int W, int C, int bs, const float* x, float* out) {
    if (B < 1 || H < 1 || W < 1 || C < 1 || bs < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    if (H % bs || W % bs) return fail(nullptr, SR_ERR_SHAPE, "space_to_depth: H and W must be multiples of the block size");
    const size_t n = (size_t)B * H * W * C;
    return simple_op(device, n, n, x, out, [&](const float* a, float* o) {
        return launch_s2d_generic(a, o, B, H, W, C, bs, nullptr);
    });
}

int sr_op_blur_cubic_downscale(int device, int N, int H, int W, int C, const float* hr, float* lr) {
    if (N < 1 || H < 4 || W < 4 || C < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    if (H % 4 || W % 4) return fail(nullptr, SR_ERR_SHAPE, "blur + cubic x1/4: H and W must be multiples of 4");
    const size_t n_in = (size_t)N * H * W * C;
    return simple_op(device, n_in, n_in / 16, hr, lr, [&](const float* a, float* o) {
        return launch_blur_cubic_down4(a, o, N, H, W, C, nullptr);
    });
}

int sr_op_cubic_upscale4(int device, int N, int h, int w, int C, const float* lr, float* out) {
    if (N < 1 || h < 1 || w < 1 || C < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    const size_t n_in = (size_t)N * h * w * C;
    return simple_op(device, n_in, n_in * 16, lr, out, [&](const float* a, float* o) { return launch_cubic_up4(a, o, N, h, w, C, nullptr); });
}

int sr_bench_conv(int device, int dtype, int transposed, int B, int H, int W, int Cin, int Cout, int residual,
                  int iters, float* ms_per_launch) {
    int rc = op_prologue(device);
    if (rc) return rc;
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || iters < 1 || !ms_per_launch)
        return fail(nullptr, SR_ERR_ARG, "bad argument");
    if (dtype == SR_F16MX6) {   // one residual-block conv on conv_trunk6.hip, mx6 in and out, random data
        if (transposed || Cin != 64 || Cout != 64) return fail(nullptr, SR_ERR_ARG, "f16mx6: the 64 -> 64 conv only");
        const size_t npix = (size_t)B * H * W, bytes = npix * 256;
        if ((double)bytes + 4096.0 > 4294967295.0) return fail(nullptr, SR_ERR_ARG, "tensor too large for 32-bit offsets");
        DevBufs d;
        float* d_f = (float*)d.get(npix * 64 * 4);
        unsigned char* bufs[2] = {(unsigned char*)d.get(bytes + 4096), (unsigned char*)d.get(bytes + 4096)};
        unsigned char* d_w = (unsigned char*)d.get(trunk6_weight_bytes());
        float* d_b = (float*)d.get(64 * 4);
        if (!d_f || !bufs[0] || !bufs[1] || !d_w || !d_b) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
        std::vector<float> w((size_t)9 * 64 * 64), bias(64, 0.01f), x(npix * 64);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        for (auto& v : w) v = rnd() * 0.1f;
        for (auto& v : x) v = rnd();
        std::vector<unsigned char> packed(trunk6_weight_bytes());
        trunk6_pack_weights(w.data(), packed.data());
        HIPCHK(nullptr, hipMemcpy(d_w, packed.data(), packed.size(), hipMemcpyHostToDevice));
        HIPCHK(nullptr, hipMemcpy(d_b, bias.data(), 64 * 4, hipMemcpyHostToDevice));
        HIPCHK(nullptr, hipMemcpy(d_f, x.data(), npix * 64 * 4, hipMemcpyHostToDevice));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(nullptr, hipMemset(bufs[k] + bytes, 0, 4096));
            LAUNCHCHK(nullptr, "nhwc_to_mx6", launch_nhwc_to_mx6(d_f, bufs[k], B, H, W, nullptr));
        }
        TrunkArgs ta = {bufs[0], bufs[1], residual ? bufs[1] : nullptr, d_w, d_b, (unsigned)bytes, 0u, B, H, W, residual ? 0 : 1, 0, 0, 0};
        if (const char* e = getenv("SR_CONV_DBG")) ta.dbg = atoi(e);      // ablations (conv_trunk6.hip), residual layer only
        hipEvent_t e0, e1;
        HIPCHK(nullptr, hipEventCreate(&e0));
        HIPCHK(nullptr, hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) LAUNCHCHK(nullptr, "trunk6", launch_trunk6_conv(ta, 0, nullptr));
        HIPCHK(nullptr, hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) LAUNCHCHK(nullptr, "trunk6", launch_trunk6_conv(ta, 0, nullptr));
        HIPCHK(nullptr, hipEventRecord(e1, nullptr));
        HIPCHK(nullptr, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(nullptr, hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_launch = ms / iters;
        return SR_OK;
    }
    if (dtype < 0 || dtype > 2) return fail(nullptr, SR_ERR_ARG, "bad dtype");
    const int mode = transposed ? M_CONVT : M_CONV;
    ConvLayer L;
    plan_layer(L, dtype, mode, Cin, Cout);
    struct Guard { ConvLayer& l; ~Guard() { free_layer(l); g_conv_dbg = 0; g_conv_stamps = nullptr; } } guard{L};
    if (const char* e = getenv("SR_CONV_DBG")) g_conv_dbg = atoi(e);
    std::vector<float> w((size_t)9 * Cin * Cout), bias(Cout, 0.01f);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : w) v = rnd() * 0.1f;
    if ((rc = upload_layer(nullptr, L, dtype, w.data(), bias.data()))) return rc;
    DevBufs d;
    const size_t npix = (size_t)B * H * W, opix = mode == M_CONV ? npix : 4 * npix;
    if (getenv("SR_BENCH_TRUNK") && dtype == SR_BF16X3 && Cin == 64 && Cout == 64) {
        // the split-planar trunk kernels (conv_trunk.hip) on random bf16 pairs; ablations via SR_CONV_DBG
        L.ents2 = transposed ? trunkT_entries(64, 64) : trunk_entries(64, 64);
        if ((rc = upload_alt(nullptr, L, trunk_weight_bytes(), w.data()))) return rc;
        const size_t bytes = npix * 256, obytes = opix * 256;
        unsigned char* bufs[2] = {(unsigned char*)d.get(bytes + 4096), (unsigned char*)d.get(obytes + 4096)};
        if (!bufs[0] || !bufs[1]) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
        std::vector<uint16_t> hb((bytes + 4096) / 2);
        for (size_t i = 0; i < bytes / 2; ++i) hb[i] = f2bf(rnd() * ((i / 8) & 1 ? 0.004f : 1.f));   // units alternate hi | lo
        HIPCHK(nullptr, hipMemcpy(bufs[0], hb.data(), bytes + 4096, hipMemcpyHostToDevice));
        if (!transposed) HIPCHK(nullptr, hipMemcpy(bufs[1], hb.data(), bytes + 4096, hipMemcpyHostToDevice));
        TrunkArgs ta = {bufs[0], bufs[1], residual && !transposed ? bufs[1] : nullptr, (const unsigned char*)L.d_w2, L.d_bias, (unsigned)bytes,
                        (unsigned)obytes, B, H, W, residual && !transposed ? 0 : 1, 0, 0, g_conv_dbg};
        auto go = [&]() { return transposed ? launch_trunk_convT(ta, nullptr) : launch_trunk_conv(ta, nullptr); };
        hipEvent_t e0, e1;
        HIPCHK(nullptr, hipEventCreate(&e0));
        HIPCHK(nullptr, hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) LAUNCHCHK(nullptr, "trunk", go());
        HIPCHK(nullptr, hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) LAUNCHCHK(nullptr, "trunk", go());
        HIPCHK(nullptr, hipEventRecord(e1, nullptr));
        HIPCHK(nullptr, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(nullptr, hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_launch = ms / iters;
        return SR_OK;
    }
    const size_t es = dtype == SR_BF16 ? 2 : 4;
    const bool out_f32 = dtype != SR_BF16 || L.tapsn;   // image / flow outputs are always fp32
    const int cout_p = L.tapsn ? Cout : round_up(Cout, 16);
    std::vector<float> x(npix * Cin);
    for (auto& v : x) v = rnd();
    float* d_x = (float*)d.get(npix * Cin * 4);
    void* d_xp = d.get(npix * L.cin_p * es);
    void* d_out = d.get(opix * cout_p * 4);
    if (!d_x || !d_xp || !d_out) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
    HIPCHK(nullptr, hipMemcpy(d_x, x.data(), npix * Cin * 4, hipMemcpyHostToDevice));
    LAUNCHCHK(nullptr, "pad_channels", launch_pad_channels(d_x, d_xp, dtype == SR_BF16, (long)npix, Cin, L.cin_p, nullptr));
    HIPCHK(nullptr, hipMemset(d_out, 0, opix * cout_p * 4));
    ConvIO io{d_xp, H, W, d_out, cout_p, Cout, out_f32 ? 1 : 0, residual ? d_out : nullptr, cout_p, A_RELU};
    if (getenv("SR_BENCH_PLANAR") && mode == M_CONV && !L.tapsn && L.steps == prec_maxsteps(dtype, mode) &&
        cout_p == Cout && (Cout * es) % 64 == 0)
        io.in_planar = io.out_planar = 1;   // the SRNet trunk's production layout
    if (getenv("SR_BENCH_PROJ") && mode == M_CONVT && Cout == 64) {   // the production form of conv2d_transpose_2: fused 64 x 27 tap projection
        std::vector<float> pw((size_t)9 * 64 * 3);
        for (auto& v : pw) v = rnd() * 0.1f;
        if ((rc = upload_proj(nullptr, L, dtype, pw.data(), 3))) return rc;
        io.out_cs = io.cout_store = 32;
        io.out_f32 = 1;
        io.proj_w = L.d_proj;
        if (getenv("SR_BENCH_GATHER")) {   // ... and the in-kernel tap sum: image out, d_out takes the border partial sums
            io.gather_out = (float*)d.get(opix * 3 * 4);
            if (!io.gather_out) return fail(nullptr, SR_ERR_HIP, "hipMalloc failed");
            io.gather_bias = L.d_bias;
            io.gather_cs = io.gather_cout = 3;
            io.gather_act = A_CLIP01;
        }
    }
    for (int i = 0; i < 3; ++i)
        if ((rc = run_conv(nullptr, dtype, nullptr, L, B, io, "bench_conv"))) return rc;
    if (g_conv_dbg & 64) {
        g_conv_stamps = (unsigned long long*)d.get(16 * 8);
        HIPCHK(nullptr, hipMemset(g_conv_stamps, 0, 16 * 8));
    }
    hipEvent_t e0, e1;
    HIPCHK(nullptr, hipEventCreate(&e0));
    HIPCHK(nullptr, hipEventCreate(&e1));
    HIPCHK(nullptr, hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i)
        if ((rc = run_conv(nullptr, dtype, nullptr, L, B, io, "bench_conv"))) return rc;
    HIPCHK(nullptr, hipEventRecord(e1, nullptr));
    HIPCHK(nullptr, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(nullptr, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / iters;
    if (g_conv_stamps) {
        unsigned long long hs[16];
        HIPCHK(nullptr, hipMemcpy(hs, g_conv_stamps, sizeof hs, hipMemcpyDeviceToHost));
        g_conv_stamps = nullptr;
        static const char* nm[8] = {"fetch_issue", "mfma_loop", "barrier1", "lds_store", "barrier2", "epilogue", "tile_total", "n"};
        double tot = (double)hs[6];
        for (int i = 0; i < 7; ++i) fprintf(stderr, "  stamp %-12s %6.1f %%  (%.0f cycles / tile / wave0)\n", nm[i], 100.0 * hs[i] / tot, (double)hs[i] / (double)hs[7]);
        static const char* en[4] = {"epi.res_issue", "epi.lds_write", "epi.first_rd", "epi.add_store"};
        for (int i = 0; i < 4; ++i) fprintf(stderr, "  stamp %-14s (%.0f cycles / tile / wave0)\n", en[i], (double)hs[8 + i] / (double)hs[7]);
    }
    return SR_OK;
}

}  // extern "C"
