// 3x3 convolution with at most 3 output channels (SRNet's last layer 64->3, reference
// layers.py:75; FNet's flow head 32->2, layers.py:44) with the TAPS folded into the GEMM's N dimension.
//
// out[y,x,co] = b[co] + sum_tap P[y+ky-1, x+kx-1][tap,co],   P[p][tap,co] = sum_ci in[p,ci] * w[tap,ci,co]
//
// P is ONE GEMM per input pixel with N = 9 taps x 3 channels = 27 <= 32 columns and K = cin, so every
// staged input pixel meets the matrix cores once (4 k-steps for 64 channels) instead of once per tap
// with 29 of 32 MFMA columns wasted: 9x fewer MFMAs and LDS fragment reads than the generic kernel.
// The kernel is then bound by streaming the input (256 B per HR pixel for the 64->3 layer) from HBM.
//
// One workgroup (256 threads) per 8x32 output tile, persistent over an XCD-contiguous band of tiles:
//   1. the 10x34 halo tile of one 128-byte channel chunk (a full cache line per pixel) goes
//      global -> registers (prefetched one chunk / tile ahead) -> LDS, pixel major + 16 B pad;
//   2. each wave runs the pixel GEMM over its 32-pixel groups of the flat halo tile (weights live in
//      registers for the whole kernel), accumulating over the chunks;
//   3. P goes to LDS (33-float rows, conflict free), every thread gathers the 27 partial sums of its
//      output pixel, adds bias, activation (+ residual) and stores cout floats.
#include "kernels.h"

#include <algorithm>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

template <int PREC>
struct TG {
    static constexpr int TH = 8, HROWS = TH + 2, HCOLS = TW + 2;
    static constexpr int NPIX = HROWS * HCOLS;                 // 340 halo pixels
    static constexpr int MTILES = (NPIX + 31) / 32;            // 11 groups of 32 flat pixels
    static constexpr int NPIXP = MTILES * 32;
    static constexpr int ESZ = PREC == P_BF16 ? 2 : 4;
    static constexpr int CK = tapsn_chunk(PREC);               // channels per chunk: 128 B per pixel
    static constexpr int KS = prec_ks(PREC);
    static constexpr int NSTEP = CK / KS;                      // fp32: 4, bf16: 4, bf16x3: 2
    static constexpr int PLANES = PREC == P_BF16X3 ? 2 : 1;    // hi, lo
    static constexpr int PS = 32 * NSTEP + 16;                 // LDS bytes per pixel per plane
    static constexpr int PLANE_BYTES = NPIXP * PS;
    static constexpr int A_BYTES = PLANES * PLANE_BYTES;
    static constexpr int NV = NPIX * 8;                        // 16-B global vectors per chunk
    static constexpr int ITER = (NV + 255) / 256;
    static constexpr int PROW = 33;                            // floats per P row (odd: conflict free)
    static constexpr int P_BYTES = NPIX * PROW * 4;
    static constexpr int LDS = A_BYTES > P_BYTES ? A_BYTES : P_BYTES;
    static constexpr int MPW = (MTILES + 3) / 4;               // pixel groups per wave
};

template <int PREC>
__device__ __forceinline__ void mma_t(f32x16& acc, const u32x4& w, const u32x4& x) {
    if constexpr (PREC == P_FP32) {
        const f32x4 wv = __builtin_bit_cast(f32x4, w);
        const f32x4 xv = __builtin_bit_cast(f32x4, x);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e], xv[e], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x),
                                                      acc, 0, 0, 0);
    }
}

__device__ __forceinline__ float act_t(float v, int act) {
    switch (act) {
        case A_RELU: return fmaxf(v, 0.f);
        case A_LRELU: return fmaxf(v, 0.2f * v);
        case A_TANH: return 1.f - 2.f / (__expf(2.f * v) + 1.f);
        case A_CLIP01: return fminf(fmaxf(v, 0.f), 1.f);
        default: return v;
    }
}

template <int PREC, int NCHUNK>
__global__ __launch_bounds__(256, 2) void conv3x3_tapsn_kernel(const ConvArgs a) {
    using T = TG<PREC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;

    // XCD-contiguous bands: workgroup ids go round-robin over the 8 XCDs, so id % 8 picks the band and
    // neighbouring tiles (which share halo rows) meet in one XCD's L2
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int t = xcd * per_xcd + slot;
    if (t >= t_end) return;

    // the layer's weights: NCHUNK x NSTEP (x hi, lo) A fragments, resident in registers
    u32x4 wf[NCHUNK][T::NSTEP][T::PLANES];
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(a.wpack);
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int s = 0; s < T::NSTEP; ++s)
#pragma unroll
                for (int q = 0; q < T::PLANES; ++q) wf[c][s][q] = wp[((c * T::NSTEP + s) * T::PLANES + q) * 64 + lane];
    }

    // staging plan (tile independent)
    unsigned goff[T::ITER], loff[T::ITER];
#pragma unroll
    for (int it = 0; it < T::ITER; ++it) {
        const int v = tid + it * 256;
        const int p = v >> 3, part = v & 7;
        const int py = p / T::HCOLS, px = p - py * T::HCOLS;
        goff[it] = (unsigned)((py * W + px) * a.cin_p * T::ESZ + part * 16);
        loff[it] = v < T::NV ? (unsigned)(p * T::PS + part * (PREC == P_BF16X3 ? 8 : 16)) : ~0u;
    }
    auto tile_mask = [&](int y0, int x0) {
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < T::ITER; ++it) {
            const int v = tid + it * 256;
            const int p = v >> 3;
            const int py = p / T::HCOLS, px = p - py * T::HCOLS;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            m |= ((v < T::NV && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1u : 0u) << it;
        }
        return m;
    };
    struct TileXY {
        int b, y0, x0;
    };
    // tile order: 8x8-tile super tiles, row major inside, so that the ~64 workgroups of an XCD work on
    // a compact block at any time and halo rows / columns come from L2 instead of HBM a second time
    const int per_img = a.tiles_x * a.tiles_y;
    const int nsx = (a.tiles_x + 7) >> 3;
    auto decode = [&](int tt) {
        TileXY q;
        q.b = tt / per_img;
        int rem = tt - q.b * per_img;
        const int sy = rem / (8 * a.tiles_x);
        rem -= sy * 8 * a.tiles_x;
        const int rows = min(8, a.tiles_y - 8 * sy);
        const int sx = min(rem / (rows * 8), nsx - 1);
        rem -= sx * rows * 8;
        const int cols = min(8, a.tiles_x - 8 * sx);
        const int iy = rem / cols, ix = rem - iy * cols;
        q.y0 = (8 * sy + iy) * T::TH;
        q.x0 = (8 * sx + ix) * TW;
        return q;
    };
    const unsigned char* in0 = reinterpret_cast<const unsigned char*>(a.in);
    const size_t px_bytes = (size_t)a.cin_p * T::ESZ;
    const size_t img_bytes = (size_t)H * W * px_bytes;
    auto tile_base = [&](const TileXY& q) {
        return in0 + q.b * img_bytes + ((ptrdiff_t)(q.y0 - 1) * W + (q.x0 - 1)) * (ptrdiff_t)px_bytes;
    };
    u32x4 regs[T::ITER];
    auto fetch = [&](const unsigned char* base, unsigned mask) {
#pragma unroll
        for (int it = 0; it < T::ITER; ++it) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if ((mask >> it) & 1u) v = *reinterpret_cast<const u32x4*>(base + goff[it]);
            regs[it] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < T::ITER; ++it) {
            if (loff[it] != ~0u) {
                if constexpr (PREC == P_BF16X3) {
                    const f32x4 f = __builtin_bit_cast(f32x4, regs[it]);
                    bf16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)f[e];
                        lo[e] = (__bf16)(f[e] - (float)hi[e]);
                    }
                    *reinterpret_cast<u32x2*>(smem + loff[it]) = __builtin_bit_cast(u32x2, hi);
                    *reinterpret_cast<u32x2*>(smem + T::PLANE_BYTES + loff[it]) = __builtin_bit_cast(u32x2, lo);
                } else {
                    *reinterpret_cast<u32x4*>(smem + loff[it]) = regs[it];
                }
            }
        }
    };

    TileXY cur = decode(t);
    unsigned cur_mask = tile_mask(cur.y0, cur.x0);
    fetch(tile_base(cur), cur_mask);

    float bias[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bias[co] = a.bias[co];
    float* P = reinterpret_cast<float*>(smem);
    const int oy = tid >> 5, ox = tid & 31;

    for (;;) {
        const int tn = t + nslot;
        const bool have_next = tn < t_end;
        const TileXY nxt = decode(have_next ? tn : t);
        const unsigned nxt_mask = tile_mask(nxt.y0, nxt.x0);

        f32x16 acc[T::MPW];
#pragma unroll
        for (int i = 0; i < T::MPW; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            stash();
            __syncthreads();
            if (c + 1 < NCHUNK) fetch(tile_base(cur) + (size_t)(c + 1) * 128, cur_mask);
            else if (have_next) fetch(tile_base(nxt), nxt_mask);
#pragma unroll
            for (int i = 0; i < T::MPW; ++i) {
                const int mt = wave + 4 * i;
                if (mt < T::MTILES) {
                    const unsigned char* xp = smem + (mt * 32 + r) * T::PS + 16 * h;
#pragma unroll
                    for (int s = 0; s < T::NSTEP; ++s) {
                        const u32x4 x0v = *reinterpret_cast<const u32x4*>(xp + 32 * s);
                        if constexpr (PREC == P_BF16X3) {
                            const u32x4 x1v = *reinterpret_cast<const u32x4*>(xp + T::PLANE_BYTES + 32 * s);
                            mma_t<PREC>(acc[i], wf[c][s][0], x1v);   // w_hi * x_lo
                            mma_t<PREC>(acc[i], wf[c][s][1], x0v);   // w_lo * x_hi
                            mma_t<PREC>(acc[i], wf[c][s][0], x0v);   // w_hi * x_hi
                        } else {
                            mma_t<PREC>(acc[i], wf[c][s][0], x0v);
                        }
                    }
                }
            }
            __syncthreads();   // every wave is done with this chunk: LDS may be overwritten
        }

        // P -> LDS: lane (pixel r of the group, half h) holds columns n = 8g + 4h + j
#pragma unroll
        for (int i = 0; i < T::MPW; ++i) {
            const int mt = wave + 4 * i;
            const int pix = mt * 32 + r;
            if (mt < T::MTILES && pix < T::NPIX) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = 8 * g + 4 * h + j;
                        if (n < 27) P[pix * T::PROW + n] = acc[i][4 * g + j];
                    }
            }
        }
        __syncthreads();
        {
            float o[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float* pp = P + ((oy + ky) * T::HCOLS + ox + kx) * T::PROW + (ky * 3 + kx) * 3;
#pragma unroll
                    for (int co = 0; co < 3; ++co) o[co] += pp[co];
                }
            const int gy = cur.y0 + oy, gx = cur.x0 + ox;
            if (gy < H && gx < W) {
                const size_t pix = ((size_t)cur.b * H + gy) * W + gx;
                float* op = reinterpret_cast<float*>(a.out) + pix * a.out_cs;
                const float* rp = a.res ? reinterpret_cast<const float*>(a.res) + pix * a.res_cs : nullptr;
#pragma unroll
                for (int co = 0; co < 3; ++co)
                    if (co < a.cout_store) {
                        float v = act_t(o[co], a.act);
                        if (rp) v += rp[co];
                        op[co] = v;
                    }
            }
        }
        if (!have_next) break;
        __syncthreads();   // P has been read: the next tile's chunk may land in LDS
        t = tn;
        cur = nxt;
        cur_mask = nxt_mask;
    }
}

// Second half of the fused transposed-conv + tap projection (conv_mfma.hip epilogue_proj): the input
// already holds P[pixel][tap*3 + co] (one 128-B line per pixel); stage the 10x34 halo tile of P in LDS
// and add the nine shifted planes.  Pure streaming: 128 B in, 4*cout B out per pixel.
__global__ __launch_bounds__(256, 2) void tap_gather_kernel(const ConvArgs a) {
    using T = TG<P_FP32>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int t = xcd * per_xcd + slot;
    if (t >= t_end) return;

    unsigned goff[T::ITER], loff[T::ITER];
#pragma unroll
    for (int it = 0; it < T::ITER; ++it) {
        const int v = tid + it * 256;
        const int p = v >> 3, part = v & 7;
        const int py = p / T::HCOLS, px = p - py * T::HCOLS;
        goff[it] = (unsigned)((py * W + px) * 128 + part * 16);
        loff[it] = v < T::NV ? (unsigned)((p * T::PROW + part * 4) * 4) : ~0u;
    }
    auto tile_mask = [&](int y0, int x0) {
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < T::ITER; ++it) {
            const int v = tid + it * 256;
            const int p = v >> 3;
            const int py = p / T::HCOLS, px = p - py * T::HCOLS;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            m |= ((v < T::NV && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 1u : 0u) << it;
        }
        return m;
    };
    struct TileXY {
        int b, y0, x0;
    };
    const int per_img = a.tiles_x * a.tiles_y;
    const int nsx = (a.tiles_x + 7) >> 3;
    auto decode = [&](int tt) {   // 8x8 super tiles, as in conv3x3_tapsn_kernel
        TileXY q;
        q.b = tt / per_img;
        int rem = tt - q.b * per_img;
        const int sy = rem / (8 * a.tiles_x);
        rem -= sy * 8 * a.tiles_x;
        const int rows = min(8, a.tiles_y - 8 * sy);
        const int sx = min(rem / (rows * 8), nsx - 1);
        rem -= sx * rows * 8;
        const int cols = min(8, a.tiles_x - 8 * sx);
        const int iy = rem / cols, ix = rem - iy * cols;
        q.y0 = (8 * sy + iy) * T::TH;
        q.x0 = (8 * sx + ix) * TW;
        return q;
    };
    const unsigned char* in0 = reinterpret_cast<const unsigned char*>(a.in);
    const size_t img_bytes = (size_t)H * W * 128;
    auto tile_base = [&](const TileXY& q) {
        return in0 + q.b * img_bytes + ((ptrdiff_t)(q.y0 - 1) * W + (q.x0 - 1)) * (ptrdiff_t)128;
    };
    f32x4 regs[T::ITER];
    auto fetch = [&](const unsigned char* base, unsigned mask) {
#pragma unroll
        for (int it = 0; it < T::ITER; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((mask >> it) & 1u) v = *reinterpret_cast<const f32x4*>(base + goff[it]);
            regs[it] = v;
        }
    };
    TileXY cur = decode(t);
    fetch(tile_base(cur), tile_mask(cur.y0, cur.x0));
    float bias[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bias[co] = a.bias[co];
    float* P = reinterpret_cast<float*>(smem);
    const int oy = tid >> 5, ox = tid & 31;
    for (;;) {
        const int tn = t + nslot;
        const bool have_next = tn < t_end;
        const TileXY nxt = decode(have_next ? tn : t);
#pragma unroll
        for (int it = 0; it < T::ITER; ++it) {
            if (loff[it] != ~0u) {
                float* d = reinterpret_cast<float*>(smem + loff[it]);   // 33-float rows: scalar stores
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = regs[it][e];
            }
        }
        __syncthreads();
        if (have_next) fetch(tile_base(nxt), tile_mask(nxt.y0, nxt.x0));
        float o[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* pp = P + ((oy + ky) * T::HCOLS + ox + kx) * T::PROW + (ky * 3 + kx) * 3;
#pragma unroll
                for (int co = 0; co < 3; ++co) o[co] += pp[co];
            }
        const int gy = cur.y0 + oy, gx = cur.x0 + ox;
        if (gy < H && gx < W) {
            const size_t pix = ((size_t)cur.b * H + gy) * W + gx;
            float* op = reinterpret_cast<float*>(a.out) + pix * a.out_cs;
            const float* rp = a.res ? reinterpret_cast<const float*>(a.res) + pix * a.res_cs : nullptr;
#pragma unroll
            for (int co = 0; co < 3; ++co)
                if (co < a.cout_store) {
                    float v = act_t(o[co], a.act);
                    if (rp) v += rp[co];
                    op[co] = v;
                }
        }
        if (!have_next) break;
        __syncthreads();
        t = tn;
        cur = nxt;
    }
}

// Border pixels of the in-kernel gather (conv_mfma.hip convT3x3_lds_kernel, ConvArgs::gather_out): one block per 8 x 64 output
// tile, one thread per pixel of its outermost ring.  out = act(bias + S0 + S1 + S2 + S3): the tile's own part, then the parts of the
// horizontal, vertical and diagonal neighbour, where the pixel has them -- a fixed order, so results do not depend on scheduling.
__global__ __launch_bounds__(192) void gather_fix_kernel(const float* __restrict__ S, float* __restrict__ out, const float* __restrict__ bias,
                                                         int B, int Ho, int Wo, int tiles_x, int tiles_y, int out_cs, int cout, int act,
                                                         int th) {
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int slot = threadIdx.x;
    int ly, lx;
    // th = rows of the output tile (8: convT3x3_lds_kernel / convT3x3_gather_kernel, 16: tail_kernel)
    if (slot < 64) ly = 0, lx = slot;
    else if (slot < 128) ly = th - 1, lx = slot - 64;
    else if (slot < 126 + th) ly = slot - 127, lx = 0;
    else if (slot < 124 + 2 * th) ly = slot - (125 + th), lx = 63;
    else return;
    const int gy = th * ty + ly, gx = 64 * tx + lx;
    if (gy >= Ho || gx >= Wo) return;
    const bool hz = (lx == 0 && gx > 0) || (lx == 63 && gx + 1 < Wo);
    const bool vt = (ly == 0 && gy > 0) || (ly == th - 1 && gy + 1 < Ho);
    const size_t simg = (size_t)B * Ho * Wo * 4;
    const size_t pix = ((size_t)b * Ho + gy) * Wo + gx;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(S + pix * 4);
    float o[3] = {bias[0] + s0[0], bias[1] + s0[1], bias[2] + s0[2]};
    if (hz) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + simg + pix * 4);
        o[0] += v[0], o[1] += v[1], o[2] += v[2];
    }
    if (vt) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + 2 * simg + pix * 4);
        o[0] += v[0], o[1] += v[1], o[2] += v[2];
    }
    if (hz && vt) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + 3 * simg + pix * 4);
        o[0] += v[0], o[1] += v[1], o[2] += v[2];
    }
    float* op = out + pix * out_cs;
#pragma unroll
    for (int co = 0; co < 3; ++co)
        if (co < cout) op[co] = act_t(o[co], act);
}

int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int PREC, int NCHUNK>
hipError_t launch_tapsn(const ConvArgs& a, hipStream_t s) {
    using T = TG<PREC>;
    static bool attr_done = false;
    auto kernel = conv3x3_tapsn_kernel<PREC, NCHUNK>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    if (ntiles <= 0) return hipSuccess;
    // persistent: about two workgroups per CU, a multiple of 8 (one band per XCD), work balanced
    const int per_xcd = (ntiles + 7) / 8;
    const int max_slots = std::max(1, 2 * cu_count() / 8);
    const int rounds = (per_xcd + max_slots - 1) / max_slots;
    const int slots = (per_xcd + rounds - 1) / rounds;
    hipLaunchKernelGGL(kernel, dim3(8 * slots), dim3(256), T::LDS, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_tap_gather(const ConvArgs& a, hipStream_t s) {
    using T = TG<P_FP32>;
    if (!a.out_f32 || a.cout_store > 3) return hipErrorInvalidValue;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    if (ntiles <= 0) return hipSuccess;
    const int per_xcd = (ntiles + 7) / 8;
    const int max_slots = std::max(1, 2 * cu_count() / 8);
    const int rounds = (per_xcd + max_slots - 1) / max_slots;
    const int slots = (per_xcd + rounds - 1) / rounds;
    hipLaunchKernelGGL(tap_gather_kernel, dim3(8 * slots), dim3(256), T::P_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_gather_fix(const ConvArgs& a, hipStream_t s, int th_in) {
    if (!a.gather_out || !a.gather_bias || a.gather_cout > 3 || (th_in != 4 && th_in != 8)) return hipErrorInvalidValue;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + th_in - 1) / th_in;   // the transposed conv's tiles: th_in x 32 input pixels
    const long n = (long)tiles_x * tiles_y * a.B;
    if (n <= 0) return hipSuccess;
    if (n > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_fix_kernel, dim3((unsigned)n), dim3(192), 0, s, reinterpret_cast<const float*>(a.out), a.gather_out,
                       a.gather_bias, a.B, 2 * a.H, 2 * a.W, tiles_x, tiles_y, a.gather_cs, a.gather_cout, a.gather_act, 2 * th_in);
    return hipGetLastError();
}

hipError_t launch_conv_tapsn(int prec, int nchunk, const ConvArgs& a, hipStream_t s) {
    if (nchunk != 1 && nchunk != 2) return hipErrorInvalidValue;
    if (!a.out_f32 || a.cout_store > 3) return hipErrorInvalidValue;
    switch (prec) {
        case P_FP32: return nchunk == 1 ? launch_tapsn<P_FP32, 1>(a, s) : launch_tapsn<P_FP32, 2>(a, s);
        case P_BF16: return nchunk == 1 ? launch_tapsn<P_BF16, 1>(a, s) : launch_tapsn<P_BF16, 2>(a, s);
        case P_BF16X3: return nchunk == 1 ? launch_tapsn<P_BF16X3, 1>(a, s) : launch_tapsn<P_BF16X3, 2>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace sr
