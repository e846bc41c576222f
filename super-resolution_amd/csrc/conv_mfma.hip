// 3x3 convolution and 3x3 stride-2 transposed convolution as NHWC implicit GEMM on
// the CDNA4 matrix cores (gfx950).  Replaces the TensorFlow Conv2D /
// Conv2DBackpropInput kernels behind reference layers.py:21-24,42-44,58-61,67-75.
//
// Decomposition (one workgroup = 256 threads = 4 wave64):
//   M = 32-pixel row segments of the output tile (TH rows x 32 columns),
//   N = NT blocks of 32 output channels, K = taps x input channels.
//   The halo tile of one input-channel chunk lives in LDS, pixel major, each
//   pixel's 32*STEPS payload bytes followed by 16 B of padding so the 32 lanes of
//   an A-fragment read (ds_read_b128, one pixel per lane) fall on 16 distinct
//   16-B bank slots.  Weights are pre-packed on the host in exactly the per-lane
//   B-fragment order.  Three kernels share the geometry:
//     conv3x3_lds_kernel   (the hot one: FNet, the heads, every stride-1 layer of the training step) keeps the
//                          chunk's packed weights in LDS next to the halo tile, copied once per workgroup and chunk;
//     convT3x3_lds_kernel  copies ONE tap's weights per barrier into one of two LDS buffers;
//     conv3x3_kernel       (layers whose channels do not fit those: the fallback) streams the weights from L2 with one
//                          coalesced 1-KiB global_load_dwordx4 per (tap, k-step, N block) per wave through a register
//                          ring that runs PF k-steps ahead of the MFMAs.
//   The next chunk's halo tile is fetched into registers while the current
//   chunk computes.  The weights are the MFMA's A operand and the pixels its B
//   operand: D then has the pixel on the lane (lane & 31) and 4 consecutive output
//   channels per register quad, so the epilogue stores 16 B per lane.
//   (The SRNet trunk's 64 -> 64 layers at inference run on conv_trunk.hip, not here.)
//
//   M_CONV : 8-row tiles, one pass over 9 taps, chunks of 32*STEPS payload bytes.
//   M_CONVT: out[2i+ky, 2j+kx] += in[i,j] w[ky,kx]  =>  four output phases
//            (ky&1, kx&1) with 4 / 2 / 2 / 1 taps, run as four passes over ONE
//            staged 4-row tile (top/left halo only), one accumulator set alive.
//
// Precisions: P_FP32   v_mfma_f32_32x32x2_f32, exact fp32 products;
//             P_BF16   v_mfma_f32_32x32x16_bf16 on bf16 activations;
//             P_BF16X3 fp32 activations split hi+lo bf16 while staging, three bf16
//                      MFMAs per product (lo*hi + hi*lo + hi*hi), fp32 accumulate.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case A_RELU: return fmaxf(v, 0.f);
        case A_LRELU: return v > 0.f ? v : 0.2f * v;
        case A_TANH: return tanhf(v);
        case A_CLIP01: return fminf(fmaxf(v, 0.f), 1.f);
        default: return v;
    }
}

__device__ __forceinline__ void keep_alive(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#endif
}

template <int PREC>
__device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (PREC == P_FP32) {
        const f32x4 av = __builtin_bit_cast(f32x4, a);
        const f32x4 bv = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
}

// PL (pixel-on-lane): weights are the A operand, pixels the B operand; otherwise the reverse
template <int PREC, bool PL>
__device__ __forceinline__ void mma_wx(f32x16& acc, const u32x4& w, const u32x4& x) {
    if constexpr (PL) mma<PREC>(acc, w, x);
    else mma<PREC>(acc, x, w);
}

template <int PREC, int MODE, int NT, int STEPS, int MTC = 2>
struct Geo {
    static constexpr int MT = MODE == M_CONV ? MTC : 1;      // 32-pixel row segments (tile rows) per wave
    static constexpr int TH = 4 * MT;                         // tile rows
    static constexpr int HB = MODE == M_CONV ? 1 : 0;         // bottom / right halo
    static constexpr int HROWS = TH + 1 + HB, HCOLS = TW + 1 + HB;
    static constexpr int NPIX = HROWS * HCOLS;
    static constexpr int PS = 32 * STEPS + 16;                // LDS bytes per halo pixel
    static constexpr int TILE_BYTES = NPIX * PS;
    static constexpr int KS = prec_ks(PREC);
    static constexpr int CKC = STEPS * KS;                    // input channels per chunk
    static constexpr int ESZ = PREC == P_BF16 ? 2 : 4;        // bytes per activation element in HBM
    static constexpr int VPP = CKC * ESZ / 16;                // 16-B global vectors per pixel per chunk
    static constexpr int NV = NPIX * VPP;
    static constexpr int ITER = (NV + 255) / 256;
    static constexpr int BV = PREC == P_BF16X3 ? 2 : 1;       // B vectors per (step, N block): hi, lo
    static constexpr int AV = PREC == P_BF16X3 ? 2 : 1;       // A vectors per (step, M block): hi, lo
    static constexpr int STEPV = NT * BV * 64;                // 16-B vectors of packed weights per k-step
    static constexpr int NPASS = MODE == M_CONV ? 1 : 4;
    // weight ring: R slots, PF = R-1 steps of prefetch; R divides every pass's step count
    static constexpr int R = MODE == M_CONV ? ((PREC == P_BF16 && STEPS % 2 == 0) ? 6 : 3)
                                            : (STEPS % 4 == 0 ? 4 : (STEPS % 2 == 0 ? 2 : 1));
    static constexpr int PF = R - 1;
};

__host__ __device__ constexpr int pass_ntaps(int mode, int pass) {
    return mode == M_CONV ? 9 : (pass == 0 ? 4 : (pass == 3 ? 1 : 2));
}
// tap index ky*3+kx of the t-th tap of a pass.  Transposed conv: phase (a,b) = (ky&1, kx&1);
// pass 0 = (0,0): ky,kx in {0,2}; pass 1 = (0,1): ky in {0,2}, kx=1; pass 2 = (1,0); pass 3 = (1,1).
__host__ __device__ constexpr int pass_tap(int mode, int pass, int t) {
    if (mode == M_CONV) return t;
    if (pass == 0) return t == 0 ? 0 : (t == 1 ? 2 : (t == 2 ? 6 : 8));
    if (pass == 1) return t == 0 ? 1 : 7;
    if (pass == 2) return t == 0 ? 3 : 5;
    return 4;
}

// vector v of a staged chunk -> (halo pixel p, 16-byte piece `part` of its channel row).  Pixel major by default; pm (planar
// inputs staged whole, the transposed conv): plane major, so that consecutive threads read consecutive 16-byte units of ONE
// plane (whole 128-byte lines) instead of four pieces of each of four planes
template <class G>
__device__ __forceinline__ void vec_map(int v, int pm, int& p, int& part) {
    if (pm) {
        constexpr int PER = G::NPIX * 4;
        const int pl = v / PER, rem = v - pl * PER;
        p = rem >> 2;
        part = pl * 4 + (rem & 3);
    } else {
        p = v / G::VPP;
        part = v % G::VPP;
    }
}

template <class G, int PREC>
__device__ __forceinline__ void stage_load(u32x4 (&regs)[G::ITER], const unsigned char* inb, int c, int cin_p,
                                           int y0, int x0, int H, int W, int tid, int planar = 0, int pm = 0) {
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        const int v = tid + it * 256;
        int p, part;
        vec_map<G>(v, pm, p, part);
        const int py = p / G::HCOLS, px = p - py * G::HCOLS;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (v < G::NV && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            size_t off = (((size_t)gy * W + gx) * cin_p + (size_t)c * G::CKC) * G::ESZ + part * 16;
            if (planar) {   // planes of 64 B per pixel (kernels.h): 16-B piece q of the pixel's channel row
                const int q = c * G::VPP + part;
                off = (size_t)(q >> 2) * H * W * 64 + ((size_t)gy * W + gx) * 64 + (q & 3) * 16;
            }
            val = *reinterpret_cast<const u32x4*>(inb + off);
        }
        regs[it] = val;
    }
}

// split: the vectors are units of a split-planar tensor (conv_trunk.hip) -- 8 bf16 hi or lo parts, already in the order
// the packed weights (pack_entries kperm) expect -- and go to the hi / lo tile as they are
template <class G, int PREC>
__device__ __forceinline__ void stage_store(const u32x4 (&regs)[G::ITER], unsigned char* smem, int tid, int split = 0, int pm = 0) {
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        const int v = tid + it * 256;
        if (v < G::NV) {
            int p, part;
            vec_map<G>(v, pm, p, part);
            if constexpr (PREC == P_BF16X3) {
                if (split) {   // part = 4 * plane + unit; unit = 2 h + (0: hi, 1: lo); k-step = plane
                    const int u = part & 3;
                    *reinterpret_cast<u32x4*>(smem + (u & 1) * G::TILE_BYTES + p * G::PS + 32 * (part >> 2) + 16 * (u >> 1)) = regs[it];
                    continue;
                }
                const f32x4 f = __builtin_bit_cast(f32x4, regs[it]);
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (__bf16)f[e];
                    lo[e] = (__bf16)(f[e] - (float)hi[e]);
                }
                *reinterpret_cast<u32x2*>(smem + p * G::PS + part * 8) = __builtin_bit_cast(u32x2, hi);
                *reinterpret_cast<u32x2*>(smem + G::TILE_BYTES + p * G::PS + part * 8) = __builtin_bit_cast(u32x2, lo);
            } else {
                *reinterpret_cast<u32x4*>(smem + p * G::PS + part * 16) = regs[it];
            }
        }
    }
}

// A fragments of one k-step: pixel (row, r + column shift) of the halo tile, 16 B at 32*s + 16*h
template <class G, int MODE>
__device__ __forceinline__ void load_a(u32x4 (&af)[G::MT][G::AV], const unsigned char* smem, int tap, int s,
                                       int wave, int r, int h) {
    const int ky = tap / 3, kx = tap % 3;
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
        int hy, hx;
        if constexpr (MODE == M_CONV) {
            hy = G::MT * wave + m + ky;
            hx = r + kx;
        } else {   // out[2i+ky] += in[i] w[ky]: ky = 2 reads the pixel above / left
            hy = wave + 1 - (ky >> 1);
            hx = r + 1 - (kx >> 1);
        }
        const int aoff = (hy * G::HCOLS + hx) * G::PS + 32 * s + 16 * h;
#pragma unroll
        for (int q = 0; q < G::AV; ++q) af[m][q] = *reinterpret_cast<const u32x4*>(smem + q * G::TILE_BYTES + aoff);
    }
}

template <class G>
__device__ __forceinline__ void load_b(u32x4 (&bf)[G::STEPV / 64], const u32x4* wbase, int gstep, int last_step) {
    const int st = gstep < last_step ? gstep : last_step;   // clamp: the ring reads past the last step harmlessly
    const u32x4* p = wbase + (size_t)st * G::STEPV;
#pragma unroll
    for (int i = 0; i < G::STEPV / 64; ++i) bf[i] = p[i * 64];
}

// one pass (= all taps of one output phase) over the staged chunk
template <class G, int PREC, int MODE, int NT, int STEPS, int PASS>
__device__ __forceinline__ void run_segment(f32x16 (&acc)[G::MT][NT], u32x4 (&ring)[G::R][G::STEPV / 64],
                                            const unsigned char* smem, const u32x4* wbase, int gstep0, int last_step,
                                            int wave, int r, int h) {
    constexpr int NTAPS = pass_ntaps(MODE, PASS);
    constexpr int NSEG = NTAPS * STEPS;
    static_assert(NSEG % G::R == 0, "ring size must divide the segment length");
    u32x4 af[2][G::MT][G::AV];
    load_a<G, MODE>(af[0], smem, pass_tap(MODE, PASS, 0), 0, wave, r, h);
#pragma unroll
    for (int i = 0; i < NSEG; ++i) {
        load_b<G>(ring[(i + G::PF) % G::R], wbase, gstep0 + i + G::PF, last_step);
        if (i + 1 < NSEG)
            load_a<G, MODE>(af[(i + 1) & 1], smem, pass_tap(MODE, PASS, (i + 1) / STEPS), (i + 1) % STEPS, wave, r, h);
        const u32x4(&bf)[G::STEPV / 64] = ring[i % G::R];
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if constexpr (PREC == P_BF16X3) {
                    mma_wx<PREC, true>(acc[m][n], bf[n * 2 + 0], af[i & 1][m][1]);   // w_hi * x_lo
                    mma_wx<PREC, true>(acc[m][n], bf[n * 2 + 1], af[i & 1][m][0]);   // w_lo * x_hi
                    mma_wx<PREC, true>(acc[m][n], bf[n * 2 + 0], af[i & 1][m][0]);   // w_hi * x_hi
                } else {
                    mma_wx<PREC, true>(acc[m][n], bf[n], af[i & 1][m][0]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// tanh(x) = 1 - 2 / (exp(2x) + 1): absolute error ~1e-7, saturates correctly at +-inf
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }

// The weights are the MFMA's A operand and the pixels its B operand, so D has the pixel on the
// lane (lane & 31) and 4 consecutive output channels n*32 + 8g + 4h + (0..3) in registers
// 4g..4g+3: a lane stores 16 B (8 B for bf16) per g, and a 3-channel output row is one lane's.
template <class G, int MODE, int NT, typename OT, bool ADD_BIAS = true>
__device__ __forceinline__ void epilogue_t(const f32x16 (&acc)[G::MT][NT], const ConvArgs& a, int pass, int b, int y0,
                                           int x0, int wave, int r, int h) {
    const int H = a.H, W = a.W;
    const int OW = MODE == M_CONV ? W : 2 * W;
    const int OH = MODE == M_CONV ? H : 2 * H;
    const int act = a.act;
    const float slope = (act == A_RELU || act == A_CLIP01) ? 0.f : (act == A_LRELU ? 0.2f : 1.f);
    const float hi = act == A_CLIP01 ? 1.f : INFINITY;
    const int ix = x0 + r;
    const bool vec = (a.out_cs % 4 == 0) && (a.cout_store % 4 == 0) && (!a.res || a.res_cs % 4 == 0) && act != A_TANH;
    OT* outp = reinterpret_cast<OT*>(a.out);
    const OT* resp = reinterpret_cast<const OT*>(a.res);
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
        int iy, oy, ox;
        if constexpr (MODE == M_CONV) {
            iy = y0 + G::MT * wave + m;
            oy = iy;
            ox = ix;
        } else {
            iy = y0 + wave;
            oy = 2 * iy + (pass >> 1);
            ox = 2 * ix + (pass & 1);
        }
        const bool pok = iy < H && ix < W;
        const size_t opix = ((size_t)b * OH + oy) * OW + ox;
        OT* op = outp + opix * a.out_cs;
        const OT* rp = resp ? resp + opix * a.res_cs : nullptr;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = (blockIdx.y * NT + n) * 32 + 8 * g + 4 * h;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + co);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[m][n][4 * g + j] + (ADD_BIAS ? bias[j] : 0.f);
                    t = t > 0.f ? t : t * slope;
                    v[j] = fminf(t, hi);
                }
                if (vec) {
                    if (pok && co < a.cout_store) {
                        if constexpr (sizeof(OT) == 4) {
                            if (rp) {
                                const f32x4 rr = *reinterpret_cast<const f32x4*>(rp + co);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += rr[j];
                            }
                            f32x4 o = {v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(op + co) = o;
                        } else {
                            if (rp) {
                                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(rp + co);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)rr[j];
                            }
                            bf16x4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                            *reinterpret_cast<bf16x4*>(op + co) = o;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (pok && co + j < a.cout_store) {
                            float t = v[j];
                            if (act == A_TANH) t = tanh_fast(t);
                            if (rp) t += (float)rp[co + j];
                            op[co + j] = (OT)t;
                        }
                    }
                }
            }
        }
    }
}

template <class G, int MODE, int NT, bool PL, bool ADD_BIAS = true>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[G::MT][NT], const ConvArgs& a, int pass, int b, int y0,
                                         int x0, int wave, int r, int h) {
    static_assert(PL, "only the pixel-on-lane operand order is built (the channel-on-lane epilogue was 2x slower)");
    if (a.out_f32) epilogue_t<G, MODE, NT, float, ADD_BIAS>(acc, a, pass, b, y0, x0, wave, r, h);
    else epilogue_t<G, MODE, NT, __bf16, ADD_BIAS>(acc, a, pass, b, y0, x0, wave, r, h);
}

// Transposed conv + tap projection of the FOLLOWING 3x3 conv with <= 3 output channels (SRNet:
// conv2d_transpose_2 -> ReLU -> conv2d_36, reference layers.py:74-75).  Instead of storing the 64
// activated channels of an output pixel (256 B) this epilogue multiplies them, still in registers,
// with the next layer's weights arranged as a 64 x 27 matrix (column = tap*3 + channel) and stores the
// 27 partial sums P[pixel][tap, co] (one 128-B line); tap_gather_kernel (conv_tapsn.hip) then adds the
// nine shifted P planes.  The accumulator layout IS the MFMA B-operand layout up to a permutation of K
// (lane (pixel r, half h) holds channels n*32 + 8g + 4h + j), which the host applies to the packed
// projection weights, so no data moves between the two MFMA stages.
// bias4[n][g] = a.bias[n*32 + 8g + 4h ..+3] and pwr[k] = the lane's projection-weight vectors, loaded ONCE at kernel start: as
// global loads inside the epilogue they queue behind the weight-tap prefetch in the in-order vmcnt counter and their latency
// is exposed once per pass
template <int PREC>
struct ProjRegs {
    static constexpr int NW = PREC == P_BF16 ? 4 : 8;
    f32x4 bias4[2][4];
    u32x4 pw[NW];
};

template <class G, int PREC>
__device__ __forceinline__ void proj_regs_load(ProjRegs<PREC>& pr, const ConvArgs& a, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) pr.bias4[n][g] = *reinterpret_cast<const f32x4*>(a.bias + (blockIdx.y * 2 + n) * 32 + 8 * g + 4 * h);
    if (a.proj_w) {
        const u32x4* pw = reinterpret_cast<const u32x4*>(a.proj_w) + lane;
#pragma unroll
        for (int k = 0; k < ProjRegs<PREC>::NW; ++k) pr.pw[k] = pw[k * 64];
    } else {
#pragma unroll
        for (int k = 0; k < ProjRegs<PREC>::NW; ++k) pr.pw[k] = u32x4{0u, 0u, 0u, 0u};
    }
}

// the projection itself: p[4 g + j] = P[pixel r][8 g + 4 h + j] of this pass
template <class G, int PREC>
__device__ __forceinline__ void proj_compute(const f32x16 (&acc)[G::MT][2], const ConvArgs& a, const ProjRegs<PREC>& pr, f32x16& p) {
    static_assert(G::MT == 1, "transposed-conv tiles have one row per wave");
    const float slope = a.act == A_RELU ? 0.f : (a.act == A_LRELU ? 0.2f : 1.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int g = 2 * s + q;
                const f32x4 bias = pr.bias4[n][g];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = acc[0][n][4 * g + j] + bias[j];
                    v[4 * q + j] = fmaxf(t, slope * t);
                }
            }
            const int kidx = n * 2 + s;
            if constexpr (PREC == P_FP32) {
                const f32x4 w0 = __builtin_bit_cast(f32x4, pr.pw[kidx * 2 + 0]);
                const f32x4 w1 = __builtin_bit_cast(f32x4, pr.pw[kidx * 2 + 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) p = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], v[e], p, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) p = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], v[4 + e], p, 0, 0, 0);
            } else {
                bf16x8 xh, xl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = (__bf16)v[e];
                    xl[e] = (__bf16)(v[e] - (float)xh[e]);
                }
                if constexpr (PREC == P_BF16X3) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, pr.pw[kidx * 2 + 0]);
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, pr.pw[kidx * 2 + 1]);
                    p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, p, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, p, 0, 0, 0);
                    p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, p, 0, 0, 0);
                } else {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, pr.pw[kidx]);
                    p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, p, 0, 0, 0);
                }
            }
        }
    }
}

template <class G, int PREC>
__device__ __forceinline__ void epilogue_proj(const f32x16 (&acc)[G::MT][2], const ConvArgs& a, const ProjRegs<PREC>& pr, int pass, int b,
                                              int y0, int x0, int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    const int iy = y0 + wave, ix = x0 + r;
    f32x16 p;
    proj_compute<G, PREC>(acc, a, pr, p);
    if (iy < H && ix < W && !(a.dbg & 256)) {   // dbg 256: ablation of the micro-benchmark, no P stores
        const int oy = 2 * iy + (pass >> 1), ox = 2 * ix + (pass & 1);
        float* op = reinterpret_cast<float*>(a.out) + (((size_t)b * 2 * H + oy) * 2 * W + ox) * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o = {p[4 * g], p[4 * g + 1], p[4 * g + 2], p[4 * g + 3]};
            *reinterpret_cast<f32x4*>(op + 8 * g) = o;
        }
    }
}

// Plain epilogue of the 64-channel transposed conv (SRNet conv2d_transpose_1): bias from registers, activation as one max, 16-byte
// stores of this lane's four channels; anything else (narrow stores, residual, tanh) takes the general epilogue_t.
template <class G>
__device__ __forceinline__ void epilogue_convt(const f32x16 (&acc)[G::MT][2], const ConvArgs& a, const f32x4 (&bias4)[2][4], int pass, int b,
                                               int y0, int x0, int wave, int r, int h) {
    const bool lean = a.out_f32 && !a.res && a.cout_store == (int)(blockIdx.y * 2 + 2) * 32 && (a.out_cs & 3) == 0 &&
                      (a.act == A_RELU || a.act == A_LRELU || a.act == A_NONE);
    if (!lean) {
        epilogue<G, M_CONVT, 2, true>(acc, a, pass, b, y0, x0, wave, r, h);
        return;
    }
    const float slope = a.act == A_RELU ? 0.f : (a.act == A_LRELU ? 0.2f : 1.f);
    const int iy = y0 + wave, ix = x0 + r;
    if (iy >= a.H || ix >= a.W) return;
    const int oy = 2 * iy + (pass >> 1), ox = 2 * ix + (pass & 1);
    float* op = reinterpret_cast<float*>(a.out) + (((size_t)b * 2 * a.H + oy) * 2 * a.W + ox) * a.out_cs + blockIdx.y * 64 + 4 * h;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = acc[0][n][4 * g + j] + bias4[n][g][j];
                o[j] = fmaxf(t, slope * t);
            }
            *reinterpret_cast<f32x4*>(op + n * 32 + 8 * g) = o;
        }
}

template <class G, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[G::MT][NT]) {
#pragma unroll
    for (int m = 0; m < G::MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
}

template <int PREC, int MODE, int NT, int STEPS>
__global__ __launch_bounds__(256) void conv3x3_kernel(const ConvArgs a) {
    using G = Geo<PREC, MODE, NT, STEPS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    int tile = blockIdx.x;
    const int txi = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int tyi = tile % a.tiles_y;
    const int b = tile / a.tiles_y;
    const int x0 = txi * TW, y0 = tyi * G::TH;
    const int H = a.H, W = a.W;

    const u32x4* wbase = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)blockIdx.y * a.wstride_cb + lane;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * H * W * a.cin_p * G::ESZ;
    const int last_step = a.wstride_cb / G::STEPV - 1;

    // weight ring prologue: steps 0 .. PF-1
    u32x4 ring[G::R][G::STEPV / 64];
#pragma unroll
    for (int i = 0; i < G::PF; ++i) load_b<G>(ring[i], wbase, i, last_step);

    u32x4 regs[G::ITER];
    stage_load<G, PREC>(regs, inb, 0, a.cin_p, y0, x0, H, W, tid, a.in_planar);
    stage_store<G, PREC>(regs, smem, tid);
    __syncthreads();

    f32x16 acc[G::MT][NT];
    if constexpr (MODE == M_CONV) {
        zero_acc<G, NT>(acc);
        constexpr int NSEG = 9 * STEPS;
        for (int c = 0; c < a.nchunk; ++c) {
            const bool more = c + 1 < a.nchunk;
            if (more) stage_load<G, PREC>(regs, inb, c + 1, a.cin_p, y0, x0, H, W, tid, a.in_planar);   // lands during the MFMAs
            run_segment<G, PREC, MODE, NT, STEPS, 0>(acc, ring, smem, wbase, c * NSEG, last_step, wave, r, h);
            if (more) {
                __syncthreads();   // every wave is done reading chunk c
                stage_store<G, PREC>(regs, smem, tid);
                __syncthreads();
            }
        }
        epilogue<G, MODE, NT, true>(acc, a, 0, b, y0, x0, wave, r, h);
    } else {
        // pass-major, chunk-minor; with one chunk (every layer of this network) the tile is staged once
        int gstep = 0;
        auto restage = [&](int c) {
            __syncthreads();
            stage_load<G, PREC>(regs, inb, c, a.cin_p, y0, x0, H, W, tid, a.in_planar);
            stage_store<G, PREC>(regs, smem, tid);
            __syncthreads();
        };
#define SR_PASS(P)                                                                                              \
        zero_acc<G, NT>(acc);                                                                                   \
        for (int c = 0; c < a.nchunk; ++c) {                                                                    \
            if (a.nchunk > 1 && (P > 0 || c > 0)) restage(c);                                                   \
            run_segment<G, PREC, MODE, NT, STEPS, P>(acc, ring, smem, wbase, gstep, last_step, wave, r, h);     \
            gstep += pass_ntaps(MODE, P) * STEPS;                                                               \
        }                                                                                                       \
        if constexpr (NT == 2) {                                                                                \
            if (a.proj_w) {                                                                                     \
                ProjRegs<PREC> pr;                                                                              \
                proj_regs_load<G, PREC>(pr, a, lane);                                                           \
                epilogue_proj<G, PREC>(acc, a, pr, P, b, y0, x0, wave, lane);                                   \
            } else epilogue<G, MODE, NT, true>(acc, a, P, b, y0, x0, wave, r, h);                               \
        } else {                                                                                                \
            epilogue<G, MODE, NT, true>(acc, a, P, b, y0, x0, wave, r, h);                                      \
        }
        SR_PASS(0)
        SR_PASS(1)
        SR_PASS(2)
        SR_PASS(3)
#undef SR_PASS
    }
}

// Epilogue through LDS (M_CONV, pixel-on-lane accumulators): each wave transposes its 64 pixels x
// NT*32 channels through a private LDS scratch so that global stores (and residual loads) are 16 B
// per lane AND contiguous across lanes: one instruction covers whole pixel rows, i.e. full 128-B lines.
template <class G, int NT, typename OT>
struct EpiLds {
    static constexpr int RB = NT * 32 * (int)sizeof(OT);   // bytes of one pixel's channel row
    static constexpr int ROW = RB + 16;                     // + pad: conflict-free 16-B column writes
    static constexpr int LPR = RB / 16;                     // lanes per pixel row
    static constexpr int RPI = 64 / LPR;                    // pixel rows per wave instruction
    static constexpr int NIT = 64 / RPI;
    static constexpr int WAVE_BYTES = 64 * ROW;
};

// What the epilogue does with the tensor behind ConvArgs::res.  Inference adds it (residual); the
// training data-gradient (MASK) treats it as the stored output y of the producing activation and
// scales the gradient by that activation's derivative: 1 where y > 0, `neg` elsewhere.
template <typename OT, bool MASK>
__device__ __forceinline__ u32x4 merge_res(u32x4 val, u32x4 rr, float neg) {
    if constexpr (sizeof(OT) == 4) {
        f32x4 x = __builtin_bit_cast(f32x4, val), y = __builtin_bit_cast(f32x4, rr);
        if constexpr (MASK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = y[j] > 0.f ? x[j] : neg * x[j];
        } else {
            x += y;
        }
        return __builtin_bit_cast(u32x4, x);
    } else {
        bf16x8 x = __builtin_bit_cast(bf16x8, val), y = __builtin_bit_cast(bf16x8, rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (MASK) x[j] = (float)y[j] > 0.f ? x[j] : (__bf16)(neg * (float)x[j]);
            else x[j] = (__bf16)((float)x[j] + (float)y[j]);
        }
        return __builtin_bit_cast(u32x4, x);
    }
}

template <class G, int NT, typename OT, bool ADD_BIAS = true, bool MASK = false>
__device__ __forceinline__ void epilogue_lds(const f32x16 (&acc)[G::MT][NT], const ConvArgs& a, int b, int y0, int x0,
                                             int wave, int lane, unsigned char* smem) {
    using E = EpiLds<G, NT, OT>;
    static_assert(G::MT % 2 == 0, "rows are transposed two at a time");
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    const int act = a.act;
    // MASK: a.act names the activation whose derivative scales the result, the conv itself has none
    const float neg = act == A_LRELU ? 0.2f : 0.f;
    const float slope = MASK ? 1.f : ((act == A_RELU || act == A_CLIP01) ? 0.f : (act == A_LRELU ? 0.2f : 1.f));
    const float hi = (!MASK && act == A_CLIP01) ? 1.f : INFINITY;
    unsigned char* sc = smem + wave * E::WAVE_BYTES;
    const int cbase = blockIdx.y * NT * 32;
    // row/column of this lane in the transposed phase
    const int prow = lane / E::LPR, chunk = lane % E::LPR;
    constexpr int EPV = 16 / (int)sizeof(OT);               // channels per 16-B vector
    const int co = cbase + chunk * EPV;
    const bool cok = co < a.cout_store;
    // byte addressing: image base + pixel * pixel stride + this lane's 16-B piece of the channel row
    constexpr int SZ = (int)sizeof(OT);
    const unsigned cob = (unsigned)(co * SZ);
    const size_t plane = (size_t)H * W * 64;
    const size_t pxs_o = a.out_planar ? 64 : (size_t)a.out_cs * SZ, pxs_r = a.out_planar ? 64 : (size_t)a.res_cs * SZ;
    const size_t lo = a.out_planar ? (size_t)(cob >> 6) * plane + (cob & 63) : cob;
    unsigned char* outp = reinterpret_cast<unsigned char*>(a.out) + (size_t)b * H * W * a.out_cs * SZ + lo;
    const unsigned char* resp = a.res ? reinterpret_cast<const unsigned char*>(a.res) + (size_t)b * H * W * a.res_cs * SZ + lo : nullptr;

#pragma unroll
  for (int mg = 0; mg < G::MT / 2; ++mg) {   // 64 pixels (two tile rows) per round through the scratch
    // 1. bias + activation, column writes: lane = pixel r, 4 channels n*32 + 8g + 4h + (0..3) per store
#pragma unroll
    for (int m = 2 * mg; m < 2 * mg + 2; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = n * 32 + 8 * g + 4 * h;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + cbase + cl);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[m][n][4 * g + j] + (ADD_BIAS ? bias[j] : 0.f);
                    t = t > 0.f ? t : t * slope;
                    v[j] = fminf(t, hi);
                }
                unsigned char* dst = sc + ((m & 1) * 32 + r) * E::ROW + cl * (int)sizeof(OT);
                if constexpr (sizeof(OT) == 4) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst) = o;
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            }
        }
    }
    const int iy0 = y0 + G::MT * wave + 2 * mg;
    if (a.pool_out) {
        // 2'. max pooling 2x2 / 2 straight from the scratch: the two tile rows of this round are one pooled row; pooled
        // pixel j takes scratch rows 2j, 2j+1 (upper) and 32+2j, 33+2j (lower).  H, W even: a window is inside or outside whole.
        const int Hp = H >> 1, Wp = W >> 1;
        unsigned char* pp = reinterpret_cast<unsigned char*>(a.pool_out) + (size_t)b * Hp * Wp * a.out_cs * SZ + cob;
        const int iyp = iy0 >> 1;
#pragma unroll
        for (int k = 0; k < (E::NIT + 3) / 4; ++k) {
            const int pj = prow + k * E::RPI;                    // 0..15
            const int ixp = (x0 >> 1) + pj;
            if (pj < 16) {
                const unsigned char* s0 = sc + (2 * pj) * E::ROW + chunk * 16;
                u32x4 v[4] = {*reinterpret_cast<const u32x4*>(s0), *reinterpret_cast<const u32x4*>(s0 + E::ROW),
                              *reinterpret_cast<const u32x4*>(s0 + 32 * E::ROW), *reinterpret_cast<const u32x4*>(s0 + 33 * E::ROW)};
                u32x4 o;
                if constexpr (SZ == 4) {
                    f32x4 m = __builtin_bit_cast(f32x4, v[0]);
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        const f32x4 t = __builtin_bit_cast(f32x4, v[q]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], t[j]);
                    }
                    o = __builtin_bit_cast(u32x4, m);
                } else {
                    bf16x8 m = __builtin_bit_cast(bf16x8, v[0]);
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        const bf16x8 t = __builtin_bit_cast(bf16x8, v[q]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) m[j] = (float)t[j] > (float)m[j] ? t[j] : m[j];
                    }
                    o = __builtin_bit_cast(u32x4, m);
                }
                if (cok && iyp < Hp && ixp < Wp) *reinterpret_cast<u32x4*>(pp + ((size_t)iyp * Wp + ixp) * a.out_cs * SZ) = o;
            }
        }
        continue;
    }
    // 2. row reads (same wave: LDS ops of one wave execute in order), residual add, coalesced stores
    const int p0 = prow;                                     // pixel of iteration 0; iteration k adds k*RPI
    const bool colok = cok;
#pragma unroll
    for (int k = 0; k < E::NIT; ++k) {
        const int p = p0 + k * E::RPI;                       // 0..63: m = p >> 5, px = p & 31
        const int iy = iy0 + (p >> 5), ix = x0 + (p & 31);
        const bool ok = colok && iy < H && ix < W;
        const size_t opix = (size_t)iy * W + ix;
        u32x4 val = *reinterpret_cast<const u32x4*>(sc + p * E::ROW + chunk * 16);
        if (ok) {
            if (resp) {
                const u32x4 rr = *reinterpret_cast<const u32x4*>(resp + opix * pxs_r);
                val = merge_res<OT, MASK>(val, rr, neg);
            }
            *reinterpret_cast<u32x4*>(outp + opix * pxs_o) = val;
        }
    }
  }
}

// Per-thread staging plan: everything about "which 16-B vector of the halo tile is mine" that does
// not depend on the tile or the chunk is computed ONCE per kernel, so staging a chunk costs one
// global_load (uniform base + 32-bit lane offset) and one ds_write per vector instead of ~50 VALU
// instructions of index arithmetic.
template <class G>
struct StagePlan {
    unsigned goff[G::ITER];   // byte offset from the halo origin (tile row -1, column -1), chunk 0
    unsigned loff[G::ITER];   // LDS byte offset in the (hi) tile; ~0u: no vector in this slot
};

template <class G, int PREC>
__device__ __forceinline__ void plan_init(StagePlan<G>& pl, int tid, int W, int px_bytes) {
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        const int v = tid + it * 256;
        const int p = v / G::VPP, part = v % G::VPP;
        const int py = p / G::HCOLS, px = p - py * G::HCOLS;
        pl.goff[it] = (unsigned)((py * W + px) * px_bytes + part * 16);
        pl.loff[it] = v < G::NV ? (unsigned)(p * G::PS + part * (PREC == P_BF16X3 ? 8 : 16)) : ~0u;
    }
}

template <class G>
__device__ __forceinline__ unsigned tile_mask(int tid, int y0, int x0, int H, int W) {
    unsigned mask = 0;
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {   // once per tile: recomputing beats holding ITER more registers
        const int v = tid + it * 256;
        const int p = v / G::VPP;
        const int py = p / G::HCOLS, px = p - py * G::HCOLS;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = v < G::NV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        mask |= (ok ? 1u : 0u) << it;
    }
    return mask;
}

template <class G>
__device__ __forceinline__ void stage_load_p(u32x4 (&regs)[G::ITER], const unsigned char* sbase, const StagePlan<G>& pl,
                                             unsigned mask) {
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        u32x4 val = {0u, 0u, 0u, 0u};
        if ((mask >> it) & 1u) val = *reinterpret_cast<const u32x4*>(sbase + pl.goff[it]);
        regs[it] = val;
    }
}

template <class G, int PREC>
__device__ __forceinline__ void stage_store_p(const u32x4 (&regs)[G::ITER], unsigned char* smem, const StagePlan<G>& pl) {
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        if (pl.loff[it] != ~0u) {
            if constexpr (PREC == P_BF16X3) {
                const f32x4 f = __builtin_bit_cast(f32x4, regs[it]);
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (__bf16)f[e];
                    lo[e] = (__bf16)(f[e] - (float)hi[e]);
                }
                *reinterpret_cast<u32x2*>(smem + pl.loff[it]) = __builtin_bit_cast(u32x2, hi);
                *reinterpret_cast<u32x2*>(smem + G::TILE_BYTES + pl.loff[it]) = __builtin_bit_cast(u32x2, lo);
            } else {
                *reinterpret_cast<u32x4*>(smem + pl.loff[it]) = regs[it];
            }
        }
    }
}

// Fast epilogue for interior tiles whose channel block is stored whole: the accumulators already
// hold the bias (they start from it), the activation is max(t, slope*t), and every address is a
// uniform base plus one per-lane offset computed once.
template <class G, int NT, typename OT, bool MASK = false>
__device__ __forceinline__ void epilogue_lds_fast(const f32x16 (&acc)[G::MT][NT], const ConvArgs& a, int b, int y0, int x0,
                                                  int wave, int lane, unsigned char* smem, unsigned long long (&es)[4], bool stamp) {
    using E = EpiLds<G, NT, OT>;
    unsigned long long e0 = stamp ? __builtin_readcyclecounter() : 0ull;
    constexpr int SZ = (int)sizeof(OT);
    const int r = lane & 31, h = lane >> 5;
    // one uniform multiplier instead of a per-element choice of activation (which compiles to a scalar
    // compare-and-branch per accumulator register: 1600 instructions for these 16 LDS stores)
    const float neg = a.act == A_LRELU ? 0.2f : 0.f;   // MASK: derivative of a.act where its output is not positive
    const float slope = MASK ? 1.f : (a.act == A_RELU ? 0.f : (a.act == A_LRELU ? 0.2f : 1.f));
    unsigned char* sc = smem + wave * E::WAVE_BYTES;
    const int prow = lane / E::LPR, chunk = lane % E::LPR;
    const int co = blockIdx.y * NT * 32 + chunk * (16 / SZ);
    unsigned char* wr = sc + r * E::ROW + 4 * h * SZ;                 // + (m&1)*32*ROW + (n*32 + 8g)*SZ
    const unsigned char* rd = sc + prow * E::ROW + chunk * 16;        // + k*RPI*ROW
    const unsigned cob = (unsigned)(co * SZ);                        // this lane's 16-B piece of the channel row
    const unsigned plane = (unsigned)(a.H * a.W * 64);
    const unsigned pxs_o = a.out_planar ? 64u : (unsigned)(a.out_cs * SZ), pxs_r = a.out_planar ? 64u : (unsigned)(a.res_cs * SZ);
    const unsigned lo = a.out_planar ? (cob >> 6) * plane + (cob & 63) : cob;
    const unsigned vo_out = prow * pxs_o + lo;
    const unsigned vo_res = prow * pxs_r + lo;
    unsigned char* outp = reinterpret_cast<unsigned char*>(a.out) + (size_t)b * a.H * a.W * a.out_cs * SZ;
    const unsigned char* resp = a.res ? reinterpret_cast<const unsigned char*>(a.res) + (size_t)b * a.H * a.W * a.res_cs * SZ : nullptr;
#pragma unroll
    for (int mg = 0; mg < G::MT / 2; ++mg) {
        const size_t pix0 = (size_t)(y0 + G::MT * wave + 2 * mg) * a.W + x0;   // uniform, within the image
        unsigned char* ob = outp + pix0 * pxs_o;
        const unsigned char* rb = resp ? resp + pix0 * pxs_r : nullptr;
        // residual first: all loads of the 64-pixel block in flight while the transpose runs
        u32x4 rr[E::NIT];
        if (rb) {
#pragma unroll
            for (int k = 0; k < E::NIT; ++k) {
                const int prel = ((k * E::RPI) >> 5) * a.W + ((k * E::RPI) & 31);          // uniform
                rr[k] = *reinterpret_cast<const u32x4*>(rb + (size_t)prel * pxs_r + vo_res);
            }
        }
        if (stamp) { unsigned long long e1 = __builtin_readcyclecounter(); es[0] += e1 - e0; e0 = e1; }
#pragma unroll
        for (int m = 2 * mg; m < 2 * mg + 2; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float t = acc[m][n][4 * g + j];
                        v[j] = fmaxf(t, slope * t);   // ReLU: slope 0, LeakyReLU: 0.2, none: 1 (max(t, t) = t)
                    }
                    unsigned char* dst = wr + (m & 1) * 32 * E::ROW + (n * 32 + 8 * g) * SZ;
                    if constexpr (SZ == 4) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(dst) = o;
                    } else {
                        bf16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                        *reinterpret_cast<bf16x4*>(dst) = o;
                    }
                }
            }
        }
        if (stamp) { unsigned long long e1 = __builtin_readcyclecounter(); es[1] += e1 - e0; e0 = e1; }
#pragma unroll
        for (int k = 0; k < E::NIT; ++k) {
            const int prel = ((k * E::RPI) >> 5) * a.W + ((k * E::RPI) & 31);              // uniform
            u32x4 val = *reinterpret_cast<const u32x4*>(rd + k * E::RPI * E::ROW);
            if (k == 1 && stamp) { unsigned long long e1 = __builtin_readcyclecounter(); es[2] += e1 - e0; e0 = e1; }
            if (rb) val = merge_res<OT, MASK>(val, rr[k], neg);
            *reinterpret_cast<u32x4*>(ob + (size_t)prel * pxs_o + vo_out) = val;
        }
        if (stamp) { unsigned long long e1 = __builtin_readcyclecounter(); es[3] += e1 - e0; e0 = e1; }
    }
}

// ---------------------------------------------------------------------------------------------
// M_CONV, LDS-resident weights: the chunk's packed weights (9 taps x STEPS k-steps, 36 KiB for two
// 32-channel N blocks) sit in LDS next to the halo tile, copied once per workgroup instead of being
// streamed through the vector L1 by each of the 4 waves, so the MFMA loop issues only LDS reads.
// The next chunk (halo tile + weights) is fetched into registers while the current one computes.
template <int PREC, int NT, int STEPS, bool PL, bool MASK = false>
__global__ __launch_bounds__(256, 2) void conv3x3_lds_kernel(const ConvArgs a) {
    using G = Geo<PREC, M_CONV, NT, STEPS>;
    constexpr int NSEG = 9 * STEPS;
    constexpr int NB = NT * G::BV;                       // weight vectors per lane per k-step
    constexpr int BVEC = NSEG * G::STEPV;                // 16-B weight vectors per chunk
    constexpr int BITER = (BVEC + 255) / 256;
    constexpr int A_BYTES = G::TILE_BYTES * G::AV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* smem_b = smem + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const size_t img_bytes = (size_t)H * W * a.cin_p * G::ESZ;

    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wpack) + (size_t)blockIdx.y * a.wstride_cb * 16;
    const unsigned char* in0 = reinterpret_cast<const unsigned char*>(a.in);
    const unsigned tid16 = (unsigned)tid * 16u;
    const size_t px_bytes = a.in_planar ? 64 : (size_t)a.cin_p * G::ESZ;
    const size_t chunk_bytes = a.in_planar ? (size_t)H * W * 64 : (size_t)G::CKC * G::ESZ;

    StagePlan<G> plan;
    plan_init<G, PREC>(plan, tid, W, (int)px_bytes);

    u32x4 regs_a[G::ITER], regs_b[BITER];
    auto load_w = [&](int c) {   // uniform base + lane offset: no per-load address arithmetic
        const unsigned char* wb = wsrc + (size_t)c * BVEC * 16;
#pragma unroll
        for (int it = 0; it < BITER; ++it) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (BVEC % 256 == 0 || tid + it * 256 < BVEC) v = *reinterpret_cast<const u32x4*>(wb + it * 4096 + tid16);
            regs_b[it] = v;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int it = 0; it < BITER; ++it)
            if (BVEC % 256 == 0 || tid + it * 256 < BVEC)
                *reinterpret_cast<u32x4*>(smem_b + it * 4096 + tid16) = regs_b[it];
    };
    struct TileXY {
        int b, y0, x0;
    };
    // tile order: 8x8-tile super tiles, row major inside; each XCD (workgroup id % 8) walks its own
    // contiguous band of that order, so the tiles in flight on one XCD form a compact block and halo
    // rows / columns are L2 hits instead of second HBM reads
    const int per_img = a.tiles_x * a.tiles_y;
    const int nsx = (a.tiles_x + 7) >> 3;
    auto decode = [&](int tt) {
        TileXY q;
        q.b = tt / per_img;
        int rem = tt - q.b * per_img;
        if (a.dbg & 32) {   // ablation: plain row-major tiles
            q.y0 = (rem / a.tiles_x) * G::TH;
            q.x0 = (rem % a.tiles_x) * TW;
            return q;
        }
        const int sy = rem / (8 * a.tiles_x);
        rem -= sy * 8 * a.tiles_x;
        const int rows = min(8, a.tiles_y - 8 * sy);
        const int sx = min(rem / (rows * 8), nsx - 1);
        rem -= sx * rows * 8;
        const int cols = min(8, a.tiles_x - 8 * sx);
        const int iy = rem / cols, ix = rem - iy * cols;
        q.y0 = (8 * sy + iy) * G::TH;
        q.x0 = (8 * sx + ix) * TW;
        return q;
    };

    // persistent over tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the (tile, chunk) sequence is one
    // software pipeline, so a tile's first chunk is fetched under the previous tile's last chunk and
    // its stores drain under the next tile's MFMAs
    const int per_xcd = (ntiles + 7) / 8;
    const bool band = !(a.dbg & 32);
    const int nslot = band ? gridDim.x >> 3 : gridDim.x;
    const int t_end = band ? min(ntiles, ((int)(blockIdx.x & 7) + 1) * per_xcd) : ntiles;
    int t = band ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (t >= t_end) return;
    TileXY cur = decode(t);
    // halo origin of a tile (row y0-1, column x0-1; never dereferenced where it lies outside the image)
    auto tile_base = [&](const TileXY& q) {
        return in0 + q.b * img_bytes + ((ptrdiff_t)(q.y0 - 1) * W + (q.x0 - 1)) * (ptrdiff_t)px_bytes;
    };
    unsigned cur_mask = tile_mask<G>(tid, cur.y0, cur.x0, H, W);
    stage_load_p<G>(regs_a, tile_base(cur), plan, cur_mask);
    load_w(0);
    stage_store_p<G, PREC>(regs_a, smem, plan);
    store_w();
    __syncthreads();

    const unsigned char* bl = smem_b + lane * 16;
    const bool stamp = (a.dbg & 64) && a.stamps;
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0}, ntile = 0, es[4] = {0, 0, 0, 0};
#define SR_T() (stamp ? __builtin_readcyclecounter() : 0ull)
    for (;;) {
        const unsigned long long t_tile = SR_T();
        f32x16 acc[G::MT][NT];
#pragma unroll
        for (int m = 0; m < G::MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
        // the accumulators start from the bias: lane (pixel r, half h) owns channels n*32 + 8g + 4h + (0..3)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + (blockIdx.y * NT + n) * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int m = 0; m < G::MT; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][n][4 * g + j] = bv[j];
            }
        const int tn = t + nslot;
        const bool have_next = tn < t_end;
        const TileXY nxt = decode(have_next ? tn : t);
        const unsigned nxt_mask = tile_mask<G>(tid, nxt.y0, nxt.x0, H, W);
        for (int c = 0; c < a.nchunk; ++c) {
            const bool last = c + 1 == a.nchunk;
            const bool pf = (!last || have_next) && !(a.dbg & 4);
            unsigned long long t0 = SR_T();
            if (pf) {   // lands while this chunk's MFMAs run
                const int cn = last ? 0 : c + 1;
                stage_load_p<G>(regs_a, tile_base(last ? nxt : cur) + (size_t)cn * chunk_bytes, plan,
                                last ? nxt_mask : cur_mask);
                load_w(cn);
            }
            unsigned long long t1 = SR_T();
            ts[0] += t1 - t0;
            if (!(a.dbg & 2)) {
                u32x4 fa[2][G::MT][G::AV], fb[2][NB];
                load_a<G, M_CONV>(fa[0], smem, 0, 0, wave, r, h);
#pragma unroll
                for (int q = 0; q < NB; ++q) fb[0][q] = *reinterpret_cast<const u32x4*>(bl + q * 1024);
#pragma unroll
                for (int i = 0; i < NSEG; ++i) {
                    if (i + 1 < NSEG) {
                        load_a<G, M_CONV>(fa[(i + 1) & 1], smem, (i + 1) / STEPS, (i + 1) % STEPS, wave, r, h);
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            fb[(i + 1) & 1][q] = *reinterpret_cast<const u32x4*>(bl + ((i + 1) * NB + q) * 1024);
                    }
#pragma unroll
                    for (int m = 0; m < G::MT; ++m) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if constexpr (PREC == P_BF16X3) {
                                mma_wx<PREC, PL>(acc[m][n], fb[i & 1][n * 2 + 0], fa[i & 1][m][1]);   // w_hi * x_lo
                                mma_wx<PREC, PL>(acc[m][n], fb[i & 1][n * 2 + 1], fa[i & 1][m][0]);   // w_lo * x_hi
                                mma_wx<PREC, PL>(acc[m][n], fb[i & 1][n * 2 + 0], fa[i & 1][m][0]);   // w_hi * x_hi
                            } else {
                                mma_wx<PREC, PL>(acc[m][n], fb[i & 1][n], fa[i & 1][m][0]);
                            }
                        }
                    }
                    // spread the next step's LDS reads between this step's MFMAs (one read after each MFMA)
                    if (i + 1 < NSEG) {
                        constexpr int NRD = G::MT * G::AV + NB;
                        constexpr int NMF = G::MT * NT * (PREC == P_BF16X3 ? 3 : (PREC == P_FP32 ? 4 : 1));
                        constexpr int NPAIR = NRD < NMF ? NRD : NMF;
#pragma unroll
                        for (int k = 0; k < NPAIR; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                        }
                        if constexpr (NMF > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NPAIR, 0);
                        if constexpr (NRD > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NPAIR, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            t0 = SR_T();
            ts[1] += t0 - t1;
            if (pf && !last) {
                __syncthreads();   // every wave is done reading this chunk
                t1 = SR_T();
                ts[2] += t1 - t0;
                stage_store_p<G, PREC>(regs_a, smem, plan);
                store_w();
                t0 = SR_T();
                ts[3] += t0 - t1;
                __syncthreads();
                t1 = SR_T();
                ts[4] += t1 - t0;
            }
        }
        unsigned long long t2 = SR_T();
        // wide, vector-aligned outputs leave through the LDS transpose; narrow ones (flow, image) directly
        const bool via_lds = PL && !(a.dbg & 8) && a.cout_store % 8 == 0 && a.out_cs % 8 == 0 &&
                             (!a.res || a.res_cs % 8 == 0) && a.act != A_TANH;
        if (a.dbg & 1) {   // ablation: keep the accumulators alive, store nothing
#pragma unroll
            for (int m = 0; m < G::MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) keep_alive(acc[m][n]);
        } else if (via_lds) {
            if constexpr (PL) {
                __syncthreads();   // the tile's last chunk has been read by every wave: LDS becomes scratch
                { unsigned long long tb = SR_T(); ts[4] += tb - t2; }
                const bool interior = cur.y0 + G::TH <= H && cur.x0 + TW <= W &&
                                      (int)(blockIdx.y + 1) * NT * 32 <= a.cout_store && a.act != A_CLIP01 && !a.pool_out;
                if (interior) {
                    if (a.out_f32) epilogue_lds_fast<G, NT, float, MASK>(acc, a, cur.b, cur.y0, cur.x0, wave, lane, smem, es, stamp);
                    else epilogue_lds_fast<G, NT, __bf16, MASK>(acc, a, cur.b, cur.y0, cur.x0, wave, lane, smem, es, stamp);
                } else {
                    if (a.out_f32) epilogue_lds<G, NT, float, false, MASK>(acc, a, cur.b, cur.y0, cur.x0, wave, lane, smem);
                    else epilogue_lds<G, NT, __bf16, false, MASK>(acc, a, cur.b, cur.y0, cur.x0, wave, lane, smem);
                }
            }
        } else {
            epilogue<G, M_CONV, NT, PL, false>(acc, a, 0, cur.b, cur.y0, cur.x0, wave, r, h);
        }
        unsigned long long t3 = SR_T();
        ts[5] += t3 - t2;
        if (have_next && !(a.dbg & 4)) {
            __syncthreads();   // scratch reads / last chunk reads are done
            t2 = SR_T();
            ts[2] += t2 - t3;
            stage_store_p<G, PREC>(regs_a, smem, plan);
            store_w();
            t3 = SR_T();
            ts[3] += t3 - t2;
            __syncthreads();
            ts[4] += SR_T() - t3;
        }
        ts[6] += SR_T() - t_tile;
        ++ntile;
        if (!have_next) break;
        t = tn;
        cur = nxt;
        cur_mask = nxt_mask;
    }
#undef SR_T
    if (stamp && tid == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(a.stamps + i, ts[i]);
        for (int i = 0; i < 4; ++i) atomicAdd(a.stamps + 8 + i, es[i]);
        atomicAdd(a.stamps + 7, ntile);
    }
}

// ---------------------------------------------------------------------------------------------
// M_CONVT with the weights shared through LDS.  conv3x3_kernel lets each of the 4 waves stream all
// weights of the layer (147 KiB for 64->64 bf16x3) from L2 through its own register ring: 4x the
// vector-L1 traffic and a ring too shallow for the L2 latency.  Here the workgroup copies one TAP's
// weights (all input channels, both N blocks: 16 KiB) into one of two LDS buffers while the previous
// tap computes: every thread moves 4 vectors per tap, and the MFMA loop reads only LDS.
// Needs the whole input-channel range in one chunk (nchunk == 1) and NT == 2.
template <int PREC, int STEPS>
__global__ __launch_bounds__(256, 2) void convT3x3_lds_kernel(const ConvArgs a) {
    using G = Geo<PREC, M_CONVT, 2, STEPS>;
    constexpr int NB = 2 * G::BV;                 // weight vectors per lane per k-step
    constexpr int TAPV = STEPS * G::STEPV;        // 16-B vectors of one tap
    static_assert(TAPV % 256 == 0, "a tap's weights split evenly over the workgroup");
    constexpr int WIT = TAPV / 256;
    constexpr int A_BYTES = G::TILE_BYTES * G::AV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* smem_b = smem + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    int tile = blockIdx.x;
    const int txi = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int tyi = tile % a.tiles_y;
    const int b = tile / a.tiles_y;
    const int x0 = txi * TW, y0 = tyi * G::TH;
    const int H = a.H, W = a.W;

    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)blockIdx.y * a.wstride_cb + tid;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * H * W * a.cin_p * G::ESZ;

    // A tap's weights go global -> LDS by DMA (global_load_lds_dwordx4: no registers, no ds_write): tap q + 1 is requested at the
    // start of tap q into the buffer tap q - 1 was read from, and awaited (vmcnt(0): nothing else is in flight in the tap loop)
    // before the barrier that ends tap q.  Taps are packed in consumption order: pass major, tap, step, n, hi/lo, lane.
    auto dma_tap = [&](int q) {
        unsigned char* dst = smem_b + (size_t)(q & 1) * TAPV * 16 + wave * 1024;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)q * TAPV + it * 256),
                                             (__attribute__((address_space(3))) void*)(dst + it * 4096), 16, 0, 0);
    };
    ProjRegs<PREC> pr;
    proj_regs_load<G, PREC>(pr, a, lane);

    if (a.dbg & 2048) {   // ablation of the micro-benchmark: no input staging (what a perfectly hidden prologue would cost)
        dma_tap(0);
    } else {
        u32x4 regs[G::ITER];
        stage_load<G, PREC>(regs, inb, 0, a.cin_p, y0, x0, H, W, tid, a.in_planar, a.in_planar);
        dma_tap(0);
        stage_store<G, PREC>(regs, smem, tid, a.in_split, a.in_planar);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[G::MT][2];
    // fused projection: the four passes' P records are kept and stored at the end of the tile (below), when the workgroup's LDS
    // (input tile + tap buffers, all free by then) holds four 8 KiB transposition areas: true for the 64-channel layers it is for
    constexpr bool DEFER = A_BYTES + 2 * TAPV * 16 >= 4 * 8192;
    f32x16 pk[DEFER ? 4 : 1];
    const unsigned char* bl = smem_b + lane * 16;
    // tap sequence: pass 0 = taps {0,2,6,8}, pass 1 = {1,7}, pass 2 = {3,5}, pass 3 = {4}
    auto tap_iter = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int pass = q < 4 ? 0 : (q < 6 ? 1 : (q < 8 ? 2 : 3));
        constexpr int tidx = q < 4 ? q : (q < 6 ? q - 4 : (q < 8 ? q - 6 : 0));
        constexpr int tap = pass_tap(M_CONVT, pass, tidx);
        constexpr bool first = tidx == 0, last = tidx + 1 == pass_ntaps(M_CONVT, pass);
        if (q + 1 < 9) dma_tap(q + 1);           // that buffer's previous contents (tap q-1) were released by the last barrier
        if (first) zero_acc<G, 2>(acc);
        const unsigned char* bq = bl + (size_t)(q & 1) * TAPV * 16;
        u32x4 fa[2][G::MT][G::AV], fb[2][NB];
        load_a<G, M_CONVT>(fa[0], smem, tap, 0, wave, r, h);
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(bq + j * 1024);
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            if (i + 1 < STEPS) {
                load_a<G, M_CONVT>(fa[(i + 1) & 1], smem, tap, i + 1, wave, r, h);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    fb[(i + 1) & 1][j] = *reinterpret_cast<const u32x4*>(bq + ((i + 1) * NB + j) * 1024);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if constexpr (PREC == P_BF16X3) {
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n * 2 + 0], fa[i & 1][0][1]);   // w_hi * x_lo
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n * 2 + 1], fa[i & 1][0][0]);   // w_lo * x_hi
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n * 2 + 0], fa[i & 1][0][0]);   // w_hi * x_hi
                } else {
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n], fa[i & 1][0][0]);
                }
            }
        }
        if (last) {
            if (a.proj_w) {
                if constexpr (DEFER) proj_compute<G, PREC>(acc, a, pr, pk[pass]);
                else epilogue_proj<G, PREC>(acc, a, pr, pass, b, y0, x0, wave, lane);
            } else epilogue_convt<G>(acc, a, pr.bias4, pass, b, y0, x0, wave, r, h);
        }
        if (q + 1 < 9) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of tap q+1 has landed
            __syncthreads();                                   // tap q+1 is in LDS for every wave; tap q's buffer is free
        }
    };
    tap_iter(std::integral_constant<int, 0>{});
    tap_iter(std::integral_constant<int, 1>{});
    tap_iter(std::integral_constant<int, 2>{});
    tap_iter(std::integral_constant<int, 3>{});
    tap_iter(std::integral_constant<int, 4>{});
    tap_iter(std::integral_constant<int, 5>{});
    tap_iter(std::integral_constant<int, 6>{});
    tap_iter(std::integral_constant<int, 7>{});
    tap_iter(std::integral_constant<int, 8>{});

    // Fused projection: the P records of the wave's two output rows leave through a per-wave LDS transposition.  A pass holds
    // the record of pixel r as 16 floats in each of lanes (r, 0) and (r, 1): stored from there, one instruction puts 32 bytes
    // into each of 32 lines (measured: 0.22 of the layer's 1.31 ms).  Instead the two passes of an output row write their 64
    // records (8 KiB: the whole row stretch of this tile) into the wave's part of the workgroup's LDS, which nobody reads any more,
    // and eight instructions store 1 KiB contiguous each.  Records 2 r and 2 r + 1 form one 256-byte LDS row of 16 pieces,
    // piece slot XOR (r & 15): writes and reads both touch all 64 banks.
    if constexpr (DEFER) if (a.proj_w && a.gather_out) {
        // In-kernel tap gather: out[y][x] = act(bias + sum over taps of P[y + ky - 1][x + kx - 1][tap]).  All P records of the 8 x 64
        // output tile go to LDS (33-float rows: the gather reads one element of 64 different records per instruction), and the
        // pixels whose nine sources lie in the tile are finished here -- their P never reaches memory (128 B per pixel written,
        // then read again by tap_gather_kernel: 3.8 GB of the 180p step's traffic).  Border pixels and the ring around the tile get
        // the partial sum over this tile's records, written to S[k]; gather_fix_kernel adds the (at most four) parts in a fixed order.
        __syncthreads();        // every wave is done with the input tile and the tap buffers
        if (a.dbg & 256) return;   // ablations of the micro-benchmark: 256 no gather phase, 512 no record writes, 1024 no global stores
        float* const PT = reinterpret_cast<float*>(smem);
        if (!(a.dbg & 512)) {
            const bool in_img = y0 + wave < H && x0 + r < W;   // records outside the image contribute nothing
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                float* rec = PT + (((2 * wave + (pass >> 1)) * 64 + 2 * r + (pass & 1)) * 33 + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (8 * g + j < 23 || (8 * g + j < 27 && h == 0)) rec[8 * g + j] = in_img ? pk[pass][4 * g + j] : 0.f;
            }
        }
        __syncthreads();
        const int Ho = 2 * H, Wo = 2 * W, oy0 = 2 * y0, ox0 = 2 * x0;
        const size_t simg = (size_t)a.B * Ho * Wo * 4;
        float* const S = reinterpret_cast<float*>(a.out);
        const float hi = a.gather_act == A_CLIP01 ? 1.f : INFINITY;
        const float lo = a.gather_act == A_NONE ? -INFINITY : 0.f;
        auto finish = [&](int ly, int lx) {
            const bool rin = (unsigned)ly < 8u, cin = (unsigned)lx < 64u;
            const bool inner = ly > 0 && ly < 7 && lx > 0 && lx < 63;
            float o[3];
#pragma unroll
            for (int co = 0; co < 3; ++co) o[co] = inner ? a.gather_bias[co] : 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sy = ly + ky - 1, sx = lx + kx - 1;
                    const bool ok = (unsigned)sy < 8u && (unsigned)sx < 64u;
                    const float* pp = PT + (ok ? (sy * 64 + sx) * 33 : 0) + (ky * 3 + kx) * 3;
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        const float v = pp[co];
                        o[co] += ok ? v : 0.f;
                    }
                }
            const int gy = oy0 + ly, gx = ox0 + lx;
            if (gy < 0 || gx < 0 || gy >= Ho || gx >= Wo) return;
            const size_t pix = ((size_t)b * Ho + gy) * Wo + gx;
            if (a.dbg & 1024) {
                asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]));
                return;
            }
            if (inner) {
                float* op = a.gather_out + pix * a.gather_cs;
#pragma unroll
                for (int co = 0; co < 3; ++co)
                    if (co < a.gather_cout) op[co] = fminf(fmaxf(o[co], lo), hi);
            } else {
                const int k = (rin ? 0 : 2) + (cin ? 0 : 1);
                *reinterpret_cast<f32x4*>(S + k * simg + pix * 4) = f32x4{o[0], o[1], o[2], 0.f};
            }
        };
        finish(tid >> 5, tid & 31);
        finish(tid >> 5, 32 + (tid & 31));
        if (tid < 66) finish(-1, tid - 1);                 // the ring around the tile: this tile's part of its neighbours' borders
        else if (tid < 132) finish(8, tid - 67);
        else if (tid < 140) finish(tid - 132, -1);
        else if (tid < 148) finish(tid - 140, 64);
    } else if (a.proj_w) {
        __syncthreads();        // every wave is done with the input tile and the tap buffers
        if (a.dbg & 256) return;   // ablation of the micro-benchmark: no P stores
        unsigned char* const scr = smem + wave * 8192;
        const int iy = y0 + wave;
        const int xlim = 2 * (W - x0);      // records of this tile inside the image
        unsigned char* const orow0 = reinterpret_cast<unsigned char*>(a.out) + (((size_t)b * 2 * H + 2 * iy) * 2 * W + 2 * x0) * 128;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& p = pk[2 * py + px];
                    const f32x4 o = {p[4 * g], p[4 * g + 1], p[4 * g + 2], p[4 * g + 3]};
                    *reinterpret_cast<f32x4*>(scr + r * 256 + (((8 * px + 2 * g + h) ^ (r & 15)) << 4)) = o;
                }
            unsigned char* const orow = orow0 + (size_t)py * 2 * W * 128;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 4 * k + (lane >> 4);
                const f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * 256 + (((lane & 15) ^ (row & 15)) << 4));
                if (iy < H && 8 * k + (lane >> 3) < xlim) *reinterpret_cast<f32x4*>(orow + k * 1024 + lane * 16) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv2d_transpose_2 + tap projection + in-kernel tap sum as a PERSISTENT kernel (bf16x3, split-planar input; the production
// form of SRNet's x4 tail).  convT3x3_lds_kernel spends a quarter of its time on the input tile: 47.5 KiB fetched, written to
// LDS and awaited before the first MFMA of every workgroup (micro-benchmark: 1.22 ms, 0.94 ms with the staging removed).  Here a
// workgroup walks tiles blockIdx.x, + gridDim.x, ...: the next tile's input is requested into registers at tap 7 (inline-asm
// loads, issued AFTER the last weight-tap DMA so that the counted wait for that tap does not include them), lands under taps 7, 8,
// the projection and is written to LDS after the gather phase; the next tile's first weight tap is DMAed at tap 8 into the tap
// buffer that is free by then.  LDS map: [tap buffer 0][input tile][tap buffer 1]; a tile with parity p reads tap q from buffer
// (q + p) & 1, its gather records (29-float rows, 58 KiB) take the input tile plus the buffer its own tap 8 was in -- the other
// buffer already holds the next tile's tap 0 -- and the next tile runs with the opposite parity.
template <int PREC, int STEPS>
__global__ __launch_bounds__(256, 2) void convT3x3_gather_kernel(const ConvArgs a) {
    using G = Geo<PREC, M_CONVT, 2, STEPS>;
    static_assert(PREC == P_BF16X3, "split-planar input");
    constexpr int NB = 2 * G::BV;
    constexpr int TAPV = STEPS * G::STEPV;
    static_assert(TAPV % 256 == 0, "a tap's weights split evenly over the workgroup");
    constexpr int WIT = TAPV / 256;
    constexpr int TAPB = TAPV * 16;
    constexpr int A_BYTES = G::TILE_BYTES * G::AV;
    constexpr int PTS = 29;                          // floats per gather record (27 used; odd: conflict-free column reads)
    static_assert(8 * 64 * PTS * 4 <= TAPB + A_BYTES, "the gather records fit in the input tile plus one tap buffer");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const smem_a = smem + TAPB;

    // raw barriers with hand-placed waits: __syncthreads() makes hipcc wait vmcnt(0), which would drain the prefetch at tap 7
    auto bar = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int H = a.H, W = a.W;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    ProjRegs<PREC> pr;
    proj_regs_load<G, PREC>(pr, a, threadIdx.x & 63);
    const size_t img_bytes = (size_t)H * W * a.cin_p * G::ESZ;
    const unsigned plane_bytes = (unsigned)(H * W * 64);
    struct TileXY {
        int b, y0, x0;
    };
    auto decode = [&](int tt) {
        TileXY q;
        const int txi = tt % a.tiles_x;
        tt /= a.tiles_x;
        q.y0 = (tt % a.tiles_y) * G::TH;
        q.b = tt / a.tiles_y;
        q.x0 = txi * TW;
        return q;
    };
    // the input tile of tile q into registers: vector v = tid + 256 it is unit (v & 3) of halo pixel (v % (4 NPIX)) >> 2 of plane
    // v / (4 NPIX); lanes without a source (outside the image, past the tile) load the image's first bytes and are zeroed on
    // the way to LDS.  Inline asm: ordinary loads could be scheduled across the DMA the counted waits rely on.
    u32x4 regs_n[G::ITER];
    unsigned okm = 0;
    auto prefetch = [&](const TileXY& q, int tid) {
        const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)q.b * img_bytes;
        okm = 0;
#pragma unroll
        for (int it = 0; it < G::ITER; ++it) {
            const int v = tid + it * 256;
            int p, part;
            vec_map<G>(v, 1, p, part);
            const int py = p / G::HCOLS, px = p - py * G::HCOLS;
            const int gy = q.y0 - 1 + py, gx = q.x0 - 1 + px;
            const bool ok = v < G::NV && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned off = ok ? (unsigned)(part >> 2) * plane_bytes + (unsigned)(gy * W + gx) * 64u + (unsigned)(part & 3) * 16u : 0u;
            okm |= ok ? (1u << it) : 0u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(regs_n[it]) : "v"(off), "s"(inb) : "memory");
        }
    };
    auto store_tile = [&](int tid) {
#pragma unroll
        for (int it = 0; it < G::ITER; ++it) {
            const int v = tid + it * 256;
            if (v < G::NV) {
                int p, part;
                vec_map<G>(v, 1, p, part);
                const int u = part & 3;
                const u32x4 z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(smem_a + (u & 1) * G::TILE_BYTES + p * G::PS + 32 * (part >> 2) + 16 * (u >> 1)) =
                    ((okm >> it) & 1u) ? regs_n[it] : z;
            }
        }
    };
    auto wait_regs = [&]() {   // every vector-memory operation is done; the registers are valid from here
        static_assert(G::ITER == 11, "register list below");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(regs_n[0]), "+v"(regs_n[1]), "+v"(regs_n[2]), "+v"(regs_n[3]), "+v"(regs_n[4]), "+v"(regs_n[5]),
                       "+v"(regs_n[6]), "+v"(regs_n[7]), "+v"(regs_n[8]), "+v"(regs_n[9]), "+v"(regs_n[10])
                     :
                     : "memory");
    };

    TileXY cur = decode(t);
    int par = 0;
    {
        const int tid0 = threadIdx.x;
        const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
        prefetch(cur, tid0);
        const u32x4* wsrc0 = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)blockIdx.y * a.wstride_cb + tid0;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc0 + it * 256),
                                             (__attribute__((address_space(3))) void*)(smem + wave0 * 1024 + it * 4096), 16, 0, 0);
        wait_regs();
        store_tile(tid0);
        bar();
    }

    for (;;) {
        const int tn = t + (int)gridDim.x;
        const bool have_next = tn < ntiles;
        const TileXY nxt = decode(have_next ? tn : t);
        // the thread index passes through an opaque copy per tile: otherwise every lane-constant address of the body is hoisted
        // out of the tile loop and the kernel spills
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int r = lane & 31, h = lane >> 5;
        const int b = cur.b, y0 = cur.y0, x0 = cur.x0;
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)blockIdx.y * a.wstride_cb + tid;
        auto tapbuf = [&](int i) { return smem + (i ? TAPB + A_BYTES : 0); };
        auto dma_tap = [&](int q, int buf) {
            unsigned char* dst = tapbuf(buf) + wave * 1024;
#pragma unroll
            for (int it = 0; it < WIT; ++it)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)q * TAPV + it * 256),
                                                 (__attribute__((address_space(3))) void*)(dst + it * 4096), 16, 0, 0);
        };
        f32x16 acc[G::MT][2];
        f32x16 pk[4];
        auto tap_iter = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int pass = q < 4 ? 0 : (q < 6 ? 1 : (q < 8 ? 2 : 3));
            constexpr int tidx = q < 4 ? q : (q < 6 ? q - 4 : (q < 8 ? q - 6 : 0));
            constexpr int tap = pass_tap(M_CONVT, pass, tidx);
            constexpr bool first = tidx == 0, last = tidx + 1 == pass_ntaps(M_CONVT, pass);
            if (q + 1 < 9) dma_tap(q + 1, (q + 1 + par) & 1);      // that buffer (tap q-1) was released by the last barrier
            else if (have_next) dma_tap(0, par ^ 1);               // tap 8: the next tile's tap 0 into the buffer tap 7 was in
            if (q == 7 && have_next) prefetch(nxt, tid);           // younger than tap 8's DMA
            if (first) zero_acc<G, 2>(acc);
            const unsigned char* bq = tapbuf((q + par) & 1) + lane * 16;
            u32x4 fa[2][G::MT][G::AV], fb[2][NB];
            load_a<G, M_CONVT>(fa[0], smem_a, tap, 0, wave, r, h);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[0][j] = *reinterpret_cast<const u32x4*>(bq + j * 1024);
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                if (i + 1 < STEPS) {
                    load_a<G, M_CONVT>(fa[(i + 1) & 1], smem_a, tap, i + 1, wave, r, h);
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        fb[(i + 1) & 1][j] = *reinterpret_cast<const u32x4*>(bq + ((i + 1) * NB + j) * 1024);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n * 2 + 0], fa[i & 1][0][1]);   // w_hi * x_lo
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n * 2 + 1], fa[i & 1][0][0]);   // w_lo * x_hi
                    mma_wx<PREC, true>(acc[0][n], fb[i & 1][n * 2 + 0], fa[i & 1][0][0]);   // w_hi * x_hi
                }
            }
            if (last) proj_compute<G, PREC>(acc, a, pr, pk[pass]);
            if (q + 1 < 9) {
                // this wave's part of tap q+1 has landed (at tap 7 the eleven younger loads of the next tile stay in flight)
                if (q == 7 && have_next) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bar();
            }
        };
        static_assert(G::ITER == 11, "the counted wait of tap 7");
        tap_iter(std::integral_constant<int, 0>{});
        tap_iter(std::integral_constant<int, 1>{});
        tap_iter(std::integral_constant<int, 2>{});
        tap_iter(std::integral_constant<int, 3>{});
        tap_iter(std::integral_constant<int, 4>{});
        tap_iter(std::integral_constant<int, 5>{});
        tap_iter(std::integral_constant<int, 6>{});
        tap_iter(std::integral_constant<int, 7>{});
        tap_iter(std::integral_constant<int, 8>{});

        // the next tile's input and first tap have landed (nothing younger is in flight yet: the gather's stores come below)
        if (have_next) wait_regs();
        bar();        // every wave is done with the input tile and with tap 8's buffer
        float* const PT = reinterpret_cast<float*>(smem + (par ? TAPB : 0));
        {
            const bool in_img = y0 + wave < H && x0 + r < W;   // records outside the image contribute nothing
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                float* rec = PT + (((2 * wave + (pass >> 1)) * 64 + 2 * r + (pass & 1)) * PTS + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (8 * g + j < 23 || (8 * g + j < 27 && h == 0)) rec[8 * g + j] = in_img ? pk[pass][4 * g + j] : 0.f;
            }
        }
        bar();
        {
            const int Ho = 2 * H, Wo = 2 * W, oy0 = 2 * y0, ox0 = 2 * x0;
            const size_t simg = (size_t)a.B * Ho * Wo * 4;
            float* const S = reinterpret_cast<float*>(a.out);
            const float hi = a.gather_act == A_CLIP01 ? 1.f : INFINITY;
            const float lo = a.gather_act == A_NONE ? -INFINITY : 0.f;
            auto finish = [&](int ly, int lx) {
                const bool rin = (unsigned)ly < 8u, cin = (unsigned)lx < 64u;
                const bool inner = ly > 0 && ly < 7 && lx > 0 && lx < 63;
                float o[3];
#pragma unroll
                for (int co = 0; co < 3; ++co) o[co] = inner ? a.gather_bias[co] : 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int sy = ly + ky - 1, sx = lx + kx - 1;
                        const bool ok = (unsigned)sy < 8u && (unsigned)sx < 64u;
                        const float* pp = PT + (ok ? (sy * 64 + sx) * PTS : 0) + (ky * 3 + kx) * 3;
#pragma unroll
                        for (int co = 0; co < 3; ++co) {
                            const float v = pp[co];
                            o[co] += ok ? v : 0.f;
                        }
                    }
                const int gy = oy0 + ly, gx = ox0 + lx;
                if (gy < 0 || gx < 0 || gy >= Ho || gx >= Wo) return;
                const size_t pix = ((size_t)b * Ho + gy) * Wo + gx;
                if (inner) {
                    float* op = a.gather_out + pix * a.gather_cs;
#pragma unroll
                    for (int co = 0; co < 3; ++co)
                        if (co < a.gather_cout) op[co] = fminf(fmaxf(o[co], lo), hi);
                } else {
                    const int k = (rin ? 0 : 2) + (cin ? 0 : 1);
                    *reinterpret_cast<f32x4*>(S + k * simg + pix * 4) = f32x4{o[0], o[1], o[2], 0.f};
                }
            };
            finish(tid >> 5, tid & 31);
            finish(tid >> 5, 32 + (tid & 31));
            if (tid < 66) finish(-1, tid - 1);
            else if (tid < 132) finish(8, tid - 67);
            else if (tid < 140) finish(tid - 132, -1);
            else if (tid < 148) finish(tid - 140, 64);
        }
        if (!have_next) break;
        bar();        // the records have been read: the input tile region takes the next tile
        store_tile(tid);
        bar();
        t = tn;
        cur = nxt;
        par ^= 1;
    }
}

size_t conv_lds_bytes(int prec, int mode, int nt, int steps, int mt) {
    const int th = mode == M_CONV ? 4 * mt : 4, hb = mode == M_CONV ? 1 : 0;
    const size_t tile = (size_t)(th + 1 + hb) * (TW + 1 + hb) * (32 * steps + 16);
    const size_t a = prec == P_BF16X3 ? 2 * tile : tile;
    if (mode == M_CONVT) return a + (nt == 2 ? (size_t)2 * steps * nt * (prec == P_BF16X3 ? 2 : 1) * 1024 : 0);   // + two tap buffers
    const size_t bv = prec == P_BF16X3 ? 2 : 1;
    const size_t ab = a + (size_t)9 * steps * nt * bv * 1024;   // + the chunk's weights
    const size_t scratch = (size_t)4 * 64 * (nt * 32 * 4 + 16);  // epilogue_lds, fp32 rows (the larger case)
    return ab > scratch ? ab : scratch;
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <typename K>
static hipError_t launch_k(K kernel, size_t lds, ConvArgs a, int th, int ncb, hipStream_t s, bool& attr_done,
                           int wg_per_cu) {
    if (lds > 64 * 1024 && !attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + th - 1) / th;
    int gx = a.tiles_x * a.tiles_y * a.B;
    if (wg_per_cu > 0) {   // persistent: one band of tiles per XCD, every workgroup the same number of tiles (+-1)
        const int per_xcd = (gx + 7) / 8;
        const int max_slots = std::max(1, wg_per_cu * num_cus() / ncb / 8);
        const int rounds = (per_xcd + max_slots - 1) / max_slots;
        gx = 8 * ((per_xcd + rounds - 1) / rounds);
    }
    hipLaunchKernelGGL(kernel, dim3(gx, ncb), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int PREC, int MODE, int NT, int STEPS>
static hipError_t launch_one(const ConvArgs& a, int ncb, hipStream_t s) {
    static bool attr_done = false;
    if (a.res_mask) {
        // the masked epilogue exists only on the LDS-transpose path of the fp32 / bf16x3 conv kernels
        if constexpr (MODE == M_CONV && PREC != P_BF16) {
            const bool ok = a.res && (a.act == A_RELU || a.act == A_LRELU) && a.cout_store % 8 == 0 &&
                            a.out_cs % 8 == 0 && a.res_cs % 8 == 0 && !(a.dbg & 8);
            if (!ok) return hipErrorInvalidValue;
            static bool attr_mask = false;
            return launch_k(&conv3x3_lds_kernel<PREC, NT, STEPS, true, true>, conv_lds_bytes(PREC, MODE, NT, STEPS, 2),
                            a, 8, ncb, s, attr_mask, 2);
        } else {
            return hipErrorInvalidValue;
        }
    }
    if constexpr (MODE == M_CONV) {
        return launch_k(&conv3x3_lds_kernel<PREC, NT, STEPS, true>, conv_lds_bytes(PREC, MODE, NT, STEPS, 2), a, 8,
                        ncb, s, attr_done, 2);
    } else {
        if constexpr (NT == 2 && (STEPS * Geo<PREC, MODE, NT, STEPS>::STEPV) % 256 == 0) {
            if (a.nchunk == 1 && !(a.dbg & 128)) {
                static bool attr2 = false;
                size_t lds = conv_lds_bytes(PREC, MODE, NT, STEPS, 1);
                if (a.gather_out) {
                    // the in-kernel gather exists in the instantiations that keep the four passes' P records (DEFER)
                    if (lds < 4 * 8192 || !a.proj_w || !a.gather_bias || a.gather_cout > 3 || ncb != 1 ||
                        (a.gather_act != A_NONE && a.gather_act != A_RELU && a.gather_act != A_CLIP01))
                        return hipErrorInvalidValue;
                    lds = std::max(lds, (size_t)GATHER_LDS);
                }
                if constexpr (PREC == P_BF16X3 && STEPS == 4) {
                    const bool no_persist = getenv("SR_NO_PERSIST_UP2") != nullptr;   // A/B switch, read per launch (two launches per step)
                    if (a.gather_out && a.in_split && a.in_planar && !no_persist && !(a.dbg & (256 | 512 | 1024 | 2048)) &&
                        (double)a.H * a.W * 256.0 < 4.0e9) {
                        static bool attr3 = false;
                        const size_t lds3 = conv_lds_bytes(PREC, MODE, NT, STEPS, 1);
                        if (!attr3) {
                            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convT3x3_gather_kernel<PREC, STEPS>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
                            if (e != hipSuccess) return e;
                            attr3 = true;
                        }
                        ConvArgs b = a;
                        b.tiles_x = (a.W + TW - 1) / TW;
                        b.tiles_y = (a.H + 3) / 4;
                        const int nt_all = b.tiles_x * b.tiles_y * b.B;
                        if (nt_all <= 0) return hipSuccess;
                        const int slots = 2 * num_cus();
                        const int rounds = (nt_all + slots - 1) / slots;
                        const int gx = (nt_all + rounds - 1) / rounds;       // every workgroup the same number of tiles (+-1)
                        hipLaunchKernelGGL((convT3x3_gather_kernel<PREC, STEPS>), dim3(gx, 1), dim3(256), lds3, s, b);
                        return hipGetLastError();
                    }
                }
                return launch_k(&convT3x3_lds_kernel<PREC, STEPS>, lds, a, 4, ncb, s, attr2, 0);
            }
        }
        if (a.gather_out) return hipErrorInvalidValue;   // only convT3x3_lds_kernel gathers
        return launch_k(&conv3x3_kernel<PREC, MODE, NT, STEPS>, conv_lds_bytes(PREC, MODE, NT, STEPS, 1), a, 4, ncb,
                        s, attr_done, 0);
    }
}

template <int PREC, int MODE, int NT>
static hipError_t launch_steps(int steps, const ConvArgs& a, int ncb, hipStream_t s) {
    constexpr int MAXS = prec_maxsteps(PREC, MODE);
    if (steps == 1) return launch_one<PREC, MODE, NT, 1>(a, ncb, s);
    if constexpr (MAXS >= 2) {
        if (steps == 2) return launch_one<PREC, MODE, NT, 2>(a, ncb, s);
    }
    if constexpr (MAXS >= 4) {
        if (steps == 4) return launch_one<PREC, MODE, NT, 4>(a, ncb, s);
    }
    if constexpr (MAXS >= 8) {
        if (steps == 8) return launch_one<PREC, MODE, NT, 8>(a, ncb, s);
    }
    return hipErrorInvalidValue;
}

template <int PREC>
static hipError_t launch_prec(int mode, int nt, int steps, const ConvArgs& a, int ncb, hipStream_t s) {
    if (mode == M_CONV) {
        if (nt == 1) return launch_steps<PREC, M_CONV, 1>(steps, a, ncb, s);
        if (nt == 2) return launch_steps<PREC, M_CONV, 2>(steps, a, ncb, s);
    } else if (nt == 2) {
        return launch_steps<PREC, M_CONVT, 2>(steps, a, ncb, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv(int prec, int mode, int nt, int steps, const ConvArgs& a, int ncb, hipStream_t s) {
    switch (prec) {
        case P_FP32: return launch_prec<P_FP32>(mode, nt, steps, a, ncb, s);
        case P_BF16: return launch_prec<P_BF16>(mode, nt, steps, a, ncb, s);
        case P_BF16X3: return launch_prec<P_BF16X3>(mode, nt, steps, a, ncb, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace sr
