// 3x3 convolution and 3x3 stride-2 transposed convolution as NHWC implicit GEMM on
// the CDNA4 matrix cores (gfx950).  Replaces the TensorFlow Conv2D /
// Conv2DBackpropInput kernels behind reference layers.py:21-24,42-44,58-61,67-75.
//
// Decomposition (one workgroup = 256 threads = 4 wave64):
//   M = 32-pixel row segments of the output tile (TH rows x 32 columns),
//   N = NT blocks of 32 output channels, K = 9 taps x input channels.
//   The (TH+2) x 34 halo tile of one input-channel chunk lives in LDS, pixel
//   major, each pixel's 32*STEPS payload bytes followed by 16 B of padding so the
//   32 lanes of an A-fragment read (ds_read_b128, one pixel per lane) fall on 16
//   distinct 16-B bank slots.  Weights are pre-packed on the host in exactly the
//   per-lane B-fragment order and stream from L2 with one coalesced 1-KiB
//   global_load_dwordx4 per (tap, k-step, N block) per wave.
//   D layout of v_mfma_*_32x32: column (lane & 31) = output channel, the 16
//   accumulator registers = pixels, so every store instruction writes 32
//   consecutive channels of one pixel per half-wave.
//
// Precisions: P_FP32   v_mfma_f32_32x32x2_f32, exact fp32 products;
//             P_BF16   v_mfma_f32_32x32x16_bf16 on bf16 activations;
//             P_BF16X3 fp32 activations split hi+lo bf16 while staging, three bf16
//                      MFMAs per product (lo*hi + hi*lo + hi*hi), fp32 accumulate.
#include "kernels.h"

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

template <int PREC, int MODE, int NT, int STEPS>
__global__ __launch_bounds__(256) void conv3x3_kernel(const ConvArgs a) {
    constexpr int MT = MODE == M_CONV ? 2 : 1;     // 32-pixel row segments per wave
    constexpr int TH = 4 * MT;                     // tile rows
    constexpr int NPH = MODE == M_CONV ? 1 : 4;    // output phases (transposed conv: 2x2)
    constexpr int NPIX = (TH + 2) * HALO_W;
    constexpr int PS = 32 * STEPS + 16;            // LDS bytes per halo pixel
    constexpr int TILE_BYTES = NPIX * PS;
    constexpr int KS = prec_ks(PREC);
    constexpr int CKC = STEPS * KS;                // input channels per chunk
    constexpr bool IN_F32 = PREC != P_BF16;
    constexpr int ESZ = IN_F32 ? 4 : 2;
    constexpr int VPP = CKC * ESZ / 16;            // 16-B global vectors per pixel per chunk
    constexpr int NV = NPIX * VPP;
    constexpr int ITER = (NV + 255) / 256;
    constexpr int BV = PREC == P_BF16X3 ? 2 : 1;   // B vectors per (step, N block): hi, lo

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
    const int x0 = txi * TW, y0 = tyi * TH;
    const int H = a.H, W = a.W;

    f32x16 acc[NPH][MT][NT];
#pragma unroll
    for (int p = 0; p < NPH; ++p)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[p][m][n][i] = 0.f;

    const u32x4* wp = reinterpret_cast<const u32x4*>(a.wpack) + (size_t)blockIdx.y * a.wstride_cb + lane;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * H * W * a.cin_p * ESZ;

    for (int c = 0; c < a.nchunk; ++c) {
        // ---- stage the halo tile of chunk c: global -> registers -> LDS ----
        u32x4 regs[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * 256;
            const int p = v / VPP, part = v % VPP;
            const int py = p / HALO_W, px = p - py * HALO_W;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (v < NV && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t off = ((size_t)gy * W + gx) * a.cin_p * ESZ + (size_t)c * CKC * ESZ + part * 16;
                val = *reinterpret_cast<const u32x4*>(inb + off);
            }
            regs[it] = val;
        }
        if (c) __syncthreads();   // every wave is done reading the previous chunk
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int v = tid + it * 256;
            if (v < NV) {
                const int p = v / VPP, part = v % VPP;
                if constexpr (PREC == P_BF16X3) {
                    const f32x4 f = __builtin_bit_cast(f32x4, regs[it]);
                    bf16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (__bf16)f[e];
                        lo[e] = (__bf16)(f[e] - (float)hi[e]);
                    }
                    *reinterpret_cast<u32x2*>(smem + p * PS + part * 8) = __builtin_bit_cast(u32x2, hi);
                    *reinterpret_cast<u32x2*>(smem + TILE_BYTES + p * PS + part * 8) = __builtin_bit_cast(u32x2, lo);
                } else {
                    *reinterpret_cast<u32x4*>(smem + p * PS + part * 16) = regs[it];
                }
            }
        }
        __syncthreads();

        // ---- 9 taps x STEPS k-steps of MFMAs ----
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int ph = MODE == M_CONV ? 0 : ((ky & 1) * 2 + (kx & 1));
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                u32x4 bf[NT][BV];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int q = 0; q < BV; ++q)
                        bf[n][q] = wp[(size_t)((((tap * STEPS + s) * NT + n) * BV + q) * 64)];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    int hy, hx;
                    if constexpr (MODE == M_CONV) {
                        hy = 2 * wave + m + ky;
                        hx = r + kx;
                    } else {   // out[2i+ky] += in[i] w[ky]: ky = 2 reads the pixel above / left
                        hy = wave + 1 - (ky >> 1);
                        hx = r + 1 - (kx >> 1);
                    }
                    const int aoff = (hy * HALO_W + hx) * PS + 32 * s + 16 * h;
                    const u32x4 af = *reinterpret_cast<const u32x4*>(smem + aoff);
                    if constexpr (PREC == P_BF16X3) {
                        const u32x4 al = *reinterpret_cast<const u32x4*>(smem + TILE_BYTES + aoff);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            mma<PREC>(acc[ph][m][n], al, bf[n][0]);   // lo * hi
                            mma<PREC>(acc[ph][m][n], af, bf[n][1]);   // hi * lo
                            mma<PREC>(acc[ph][m][n], af, bf[n][0]);   // hi * hi
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < NT; ++n) mma<PREC>(acc[ph][m][n], af, bf[n][0]);
                    }
                }
            }
        }
        wp += 9 * STEPS * NT * BV * 64;
    }

    // ---- epilogue: bias, activation, residual, store ----
    const int OW = MODE == M_CONV ? W : 2 * W;
    const int OH = MODE == M_CONV ? H : 2 * H;
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = (blockIdx.y * NT + n) * 32 + r;
                const float bias = a.bias[co];
                const bool cok = co < a.cout_store;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int px = (i & 3) + 8 * (i >> 2) + 4 * h;
                    int iy, ix, oy, ox;
                    if constexpr (MODE == M_CONV) {
                        iy = y0 + 2 * wave + m;
                        ix = x0 + px;
                        oy = iy;
                        ox = ix;
                    } else {
                        iy = y0 + wave;
                        ix = x0 + px;
                        oy = 2 * iy + (p >> 1);
                        ox = 2 * ix + (p & 1);
                    }
                    if (cok && iy < H && ix < W) {
                        const size_t opix = ((size_t)b * OH + oy) * OW + ox;
                        float v = apply_act(acc[p][m][n][i] + bias, a.act);
                        if (a.out_f32) {
                            if (a.res) v += reinterpret_cast<const float*>(a.res)[opix * a.res_cs + co];
                            reinterpret_cast<float*>(a.out)[opix * a.out_cs + co] = v;
                        } else {
                            if (a.res) v += (float)reinterpret_cast<const __bf16*>(a.res)[opix * a.res_cs + co];
                            reinterpret_cast<__bf16*>(a.out)[opix * a.out_cs + co] = (__bf16)v;
                        }
                    }
                }
            }
        }
    }
}

size_t conv_lds_bytes(int prec, int mode, int steps) {
    const int th = mode == M_CONV ? 8 : 4;
    const size_t tile = (size_t)(th + 2) * HALO_W * (32 * steps + 16);
    return prec == P_BF16X3 ? 2 * tile : tile;
}

template <int PREC, int MODE, int NT, int STEPS>
static hipError_t launch_one(const ConvArgs& a, int ncb, hipStream_t s) {
    const dim3 grid(a.tiles_x * a.tiles_y * a.B, ncb);
    const size_t lds = conv_lds_bytes(PREC, MODE, STEPS);
    hipLaunchKernelGGL((conv3x3_kernel<PREC, MODE, NT, STEPS>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int PREC, int MODE, int NT>
static hipError_t launch_steps(int steps, const ConvArgs& a, int ncb, hipStream_t s) {
    constexpr int MAXS = prec_maxsteps(PREC);
    if (steps == 1) return launch_one<PREC, MODE, NT, 1>(a, ncb, s);
    if (steps == 2) return launch_one<PREC, MODE, NT, 2>(a, ncb, s);
    if constexpr (MAXS == 4) {
        if (steps == 4) return launch_one<PREC, MODE, NT, 4>(a, ncb, s);
    }
    return hipErrorInvalidValue;
}

template <int PREC>
static hipError_t launch_prec(int mode, int nt, int steps, const ConvArgs& a, int ncb, hipStream_t s) {
    if (mode == M_CONV) {
        if (nt == 1) return launch_steps<PREC, M_CONV, 1>(steps, a, ncb, s);
        if (nt == 2) return launch_steps<PREC, M_CONV, 2>(steps, a, ncb, s);
    } else {
        if (nt == 1) return launch_steps<PREC, M_CONVT, 1>(steps, a, ncb, s);
        if (nt == 2) return launch_steps<PREC, M_CONVT, 2>(steps, a, ncb, s);
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
