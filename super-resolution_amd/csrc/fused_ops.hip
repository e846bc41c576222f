// Bandwidth-bound kernels of the generator step (gfx950): everything of
// reference FRVSR.py:58-63 / TecoGAN.py:74-76 that is not a convolution.
//   pre_srnet    bilinear x4 of the flow (layers.py:89) + dense_image_warp of the
//                previous HR estimate (layers.py:131) + space_to_depth (layers.py:109)
//                + concat with the LR frame (layers.py:66), one pass, NHWC out
//   fnet_input   concat [LR_t, LR_t-1] (layers.py:29), zero padded to 16 channels
//   maxpool2     MaxPool2D 2x2 SAME (layers.py:31,33,35)
//   resize2x     TF1-legacy bilinear x2 (layers.py:37,39,41)
//   warp3        LR warp, second model output (FRVSR.py:59)
//   teco_epilogue tanh(SRNet + bilinear x4 (LR_t)) (TecoGAN.py:74-76)
// Activations are [B,H,W,C] with C a multiple of 16, float or bf16 (template T).
#include "kernels.h"
#include "gather_math.h"

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- 16-byte activation vectors: 4 floats or 8 bf16, processed as floats ----
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ float get(int i) const { return v[i]; }
    __device__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec<__bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ float get(int i) const { return (float)v[i]; }
    __device__ void set(int i, float x) { v[i] = (__bf16)x; }
};

static inline int blocks_for(long n, int bs) { return (int)((n + bs - 1) / bs); }

// ------------------------------------------------------------------ fnet_input
template <typename T>
__global__ void fnet_input_kernel(const float* __restrict__ a, const float* __restrict__ b, T* __restrict__ out, long npix) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float v[16];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v[c] = a[p * 3 + c];
        v[3 + c] = b[p * 3 + c];
    }
#pragma unroll
    for (int c = 6; c < 16; ++c) v[c] = 0.f;
    constexpr int N = Vec<T>::N;
    Vec<T>* o = reinterpret_cast<Vec<T>*>(out + p * 16);
#pragma unroll
    for (int q = 0; q < 16 / N; ++q) {
        Vec<T> w;
#pragma unroll
        for (int i = 0; i < N; ++i) w.set(i, v[q * N + i]);
        o[q] = w;
    }
}

hipError_t launch_fnet_input(const float* lr_t, const float* lr_tm1, void* out, int act_bf16,
                             int B, int H, int W, hipStream_t s) {
    const long npix = (long)B * H * W;
    if (act_bf16)
        hipLaunchKernelGGL(fnet_input_kernel<__bf16>, dim3(blocks_for(npix, 256)), dim3(256), 0, s, lr_t, lr_tm1, (__bf16*)out, npix);
    else
        hipLaunchKernelGGL(fnet_input_kernel<float>, dim3(blocks_for(npix, 256)), dim3(256), 0, s, lr_t, lr_tm1, (float*)out, npix);
    return hipGetLastError();
}

// ------------------------------------------------------------------ srnet_input (stand-alone SRNet.predict)
template <typename T>
__global__ void srnet_input_kernel(const float* __restrict__ lr, const float* __restrict__ ss, T* __restrict__ out, long npix) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long p = idx >> 6;
    const int c = idx & 63;
    if (p >= npix) return;
    float v = 0.f;
    if (c < 3) v = lr[p * 3 + c];
    else if (c < 51) v = ss[p * 48 + c - 3];
    out[p * 64 + c] = (T)v;
}

hipError_t launch_srnet_input(const float* lr_t, const float* s_s, void* out, int act_bf16,
                              int B, int H, int W, hipStream_t s) {
    const long npix = (long)B * H * W;
    if (act_bf16)
        hipLaunchKernelGGL(srnet_input_kernel<__bf16>, dim3(blocks_for(npix * 64, 256)), dim3(256), 0, s, lr_t, s_s, (__bf16*)out, npix);
    else
        hipLaunchKernelGGL(srnet_input_kernel<float>, dim3(blocks_for(npix * 64, 256)), dim3(256), 0, s, lr_t, s_s, (float*)out, npix);
    return hipGetLastError();
}

// ------------------------------------------------------------------ maxpool2 (vectorised over channels)
template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
    constexpr int N = Vec<T>::N;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, CV = C / N;
    const long total = (long)B * Ho * Wo * CV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = idx % CV;
    long p = idx / CV;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const int y0 = 2 * oy, x0 = 2 * ox;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);   // SAME: the -inf pad never wins
    const Vec<T>* base = reinterpret_cast<const Vec<T>*>(in + (size_t)b * H * W * C) + cv;
    const Vec<T> a = base[((size_t)y0 * W + x0) * CV], bq = base[((size_t)y0 * W + x1) * CV];
    const Vec<T> c = base[((size_t)y1 * W + x0) * CV], d = base[((size_t)y1 * W + x1) * CV];
    Vec<T> o;
#pragma unroll
    for (int i = 0; i < N; ++i) o.set(i, fmaxf(fmaxf(a.get(i), bq.get(i)), fmaxf(c.get(i), d.get(i))));
    reinterpret_cast<Vec<T>*>(out)[idx] = o;
}

hipError_t launch_maxpool2(const void* in, void* out, int act_bf16, int B, int H, int W, int C, hipStream_t s) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if (act_bf16) {
        const long total = (long)B * Ho * Wo * (C / 8);
        hipLaunchKernelGGL(maxpool2_kernel<__bf16>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const __bf16*)in, (__bf16*)out, B, H, W, C);
    } else {
        const long total = (long)B * Ho * Wo * (C / 4);
        hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ resize2x (legacy bilinear, vectorised)
template <typename T>
__global__ void resize2x_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
    constexpr int N = Vec<T>::N;
    const int Ho = 2 * H, Wo = 2 * W, CV = C / N;
    const long total = (long)B * Ho * Wo * CV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = idx % CV;
    long p = idx / CV;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const int y0 = oy >> 1, x0 = ox >> 1;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = (oy & 1) * 0.5f, tx = (ox & 1) * 0.5f;
    const Vec<T>* base = reinterpret_cast<const Vec<T>*>(in + (size_t)b * H * W * C) + cv;
    const Vec<T> tl = base[((size_t)y0 * W + x0) * CV], tr = base[((size_t)y0 * W + x1) * CV];
    const Vec<T> bl = base[((size_t)y1 * W + x0) * CV], br = base[((size_t)y1 * W + x1) * CV];
    Vec<T> o;
#pragma unroll
    for (int i = 0; i < N; ++i) o.set(i, lerp2(tl.get(i), tr.get(i), bl.get(i), br.get(i), tx, ty));
    reinterpret_cast<Vec<T>*>(out)[idx] = o;
}

hipError_t launch_resize2x(const void* in, void* out, int act_bf16, int B, int H, int W, int C, hipStream_t s) {
    if (act_bf16) {
        const long total = (long)B * 4 * H * W * (C / 8);
        hipLaunchKernelGGL(resize2x_kernel<__bf16>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const __bf16*)in, (__bf16*)out, B, H, W, C);
    } else {
        const long total = (long)B * 4 * H * W * (C / 4);
        hipLaunchKernelGGL(resize2x_kernel<float>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ dense_image_warp helpers
// ------------------------------------------------------------------ pre_srnet (the fused hot-path gather)
// One workgroup = 64 consecutive LR pixels x 4 sub-rows (dy).  Thread (lp, dy) upsamples the
// flow for its 4 HR pixels, gathers/warps 4 x 3 values of the previous estimate and drops them
// at channels 3 + dy*12 .. of its pixel's 64-channel row in LDS; rows go out as 16-B vectors.
// SPLIT: the 64-channel rows leave as split-planar records (conv_trunk.hip: [B][4 planes][H][W][64 B], units of 8 bf16
// hi | lo parts in the channel order kernels.h split_perm) instead of NHWC rows.
template <typename T, bool WRITE_HR, bool SPLIT = false>
__global__ __launch_bounds__(256) void pre_srnet_kernel(const float* __restrict__ flow, const float* __restrict__ est,
                                                        const float* __restrict__ lr, T* __restrict__ out,
                                                        float* __restrict__ hr_out, int B, int H, int W) {
    __shared__ __attribute__((aligned(16))) float rows[64][64 + 4];
    // 32-bit pixel indices (the launcher refuses B*H*W >= 2^31): 64-bit divisions and address products were a third of the kernel's
    // instructions, and at 700 instructions per thread the kernel was issue-bound, not HBM-bound
    const unsigned npix = (unsigned)B * (unsigned)H * (unsigned)W;
    const int lp = threadIdx.x >> 2, dy = threadIdx.x & 3;
    const unsigned p = blockIdx.x * 64u + (unsigned)lp;
    if (p < npix) {
        const unsigned q = p / (unsigned)W;
        const int x = (int)(p - q * (unsigned)W);
        const int b = (int)(q / (unsigned)H);
        const int y = (int)(q - (unsigned)b * (unsigned)H);
        const int x1 = min(x + 1, W - 1), y1 = min(y + 1, H - 1);
        const float* fb = flow + (size_t)b * H * W * 2;
        const float2 ftl = *reinterpret_cast<const float2*>(fb + ((size_t)y * W + x) * 2);
        const float2 ftr = *reinterpret_cast<const float2*>(fb + ((size_t)y * W + x1) * 2);
        const float2 fbl = *reinterpret_cast<const float2*>(fb + ((size_t)y1 * W + x) * 2);
        const float2 fbr = *reinterpret_cast<const float2*>(fb + ((size_t)y1 * W + x1) * 2);
        const int HH = 4 * H, WW = 4 * W;
        const float* eb = est + (size_t)b * HH * WW * 3;
        const float ty = dy * 0.25f;
        const int Y = 4 * y + dy;
        float vals[12];
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const float tx = dx * 0.25f;
            const float fy = lerp2(ftl.x, ftr.x, fbl.x, fbr.x, tx, ty);   // F_HR is NOT rescaled (FRVSR.py:60)
            const float fx = lerp2(ftl.y, ftr.y, fbl.y, fbr.y, tx, ty);
            const int X = 4 * x + dx;
            const WarpCoord cy = warp_coord((float)Y - fy, HH);
            const WarpCoord cx = warp_coord((float)X - fx, WW);
            const float* r0 = eb + ((size_t)cy.f * WW + cx.f) * 3;
            const float* r1 = r0 + (size_t)WW * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                vals[dx * 3 + c] = warp_interp(r0[c], r0[3 + c], r1[c], r1[3 + c], cx.a, cy.a);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) rows[lp][3 + dy * 12 + i] = vals[i];
        if (dy == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rows[lp][c] = lr[(size_t)p * 3 + c];
        } else if (dy == 1) {
#pragma unroll
            for (int c = 51; c < 64; ++c) rows[lp][c] = 0.f;
        }
        if constexpr (WRITE_HR) {
            f32x4* o = reinterpret_cast<f32x4*>(hr_out + (((size_t)b * HH + Y) * WW + 4 * x) * 3);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 w = {vals[4 * i], vals[4 * i + 1], vals[4 * i + 2], vals[4 * i + 3]};
                o[i] = w;
            }
        }
    }
    __syncthreads();
    if constexpr (SPLIT) {
        // thread v -> (plane, row, unit): per plane the 64 rows x 4 units are 4 KiB of consecutive global bytes
        const size_t hw = (size_t)H * W;
        for (int v = threadIdx.x; v < 1024; v += 256) {
            const int plane = v >> 8, row = (v >> 2) & 63, u = v & 3;
            const unsigned pp = blockIdx.x * 64u + (unsigned)row;
            if (pp < npix) {
                const int hh = u >> 1;
                typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
                bf16x8v o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = rows[row][16 * plane + split_perm(hh, e)];
                    const __bf16 hi = (__bf16)f;
                    o[e] = (u & 1) ? (__bf16)(f - (float)hi) : hi;
                }
                const unsigned bb = pp / (unsigned)hw;
                const size_t b = bb, pix = pp - bb * (unsigned)hw;
                unsigned char* dst = reinterpret_cast<unsigned char*>(out) + ((b * 4 + plane) * hw + pix) * 64 + u * 16;
                *reinterpret_cast<bf16x8v*>(dst) = o;
            }
        }
        return;
    }
    // 64 rows x 64 channels -> global, 16 B per thread per pass, fully coalesced
    constexpr int N = Vec<T>::N;
    constexpr int VPR = 64 / N;               // vectors per row
    for (int v = threadIdx.x; v < 64 * VPR; v += 256) {
        const int row = v / VPR, part = v % VPR;
        const unsigned pp = blockIdx.x * 64u + (unsigned)row;
        if (pp < npix) {
            Vec<T> w;
#pragma unroll
            for (int i = 0; i < N; ++i) w.set(i, rows[row][part * N + i]);
            reinterpret_cast<Vec<T>*>(out + (size_t)pp * 64)[part] = w;
        }
    }
}

hipError_t launch_pre_srnet(const float* flow, const float* est_prev, const float* lr_t, void* out,
                            float* hr_warp_out, int act_bf16, int B, int H, int W, hipStream_t s) {
    const long npix = (long)B * H * W;
    if (npix >= 0x7fffffffL) return hipErrorInvalidValue;      // the kernel indexes pixels with 32 bits
    const dim3 grid(blocks_for(npix, 64)), block(256);
    if (act_bf16 == 2) {
        if (hr_warp_out) hipLaunchKernelGGL((pre_srnet_kernel<float, true, true>), grid, block, 0, s, flow, est_prev, lr_t, (float*)out, hr_warp_out, B, H, W);
        else hipLaunchKernelGGL((pre_srnet_kernel<float, false, true>), grid, block, 0, s, flow, est_prev, lr_t, (float*)out, hr_warp_out, B, H, W);
    } else if (act_bf16) {
        if (hr_warp_out) hipLaunchKernelGGL((pre_srnet_kernel<__bf16, true>), grid, block, 0, s, flow, est_prev, lr_t, (__bf16*)out, hr_warp_out, B, H, W);
        else hipLaunchKernelGGL((pre_srnet_kernel<__bf16, false>), grid, block, 0, s, flow, est_prev, lr_t, (__bf16*)out, hr_warp_out, B, H, W);
    } else {
        if (hr_warp_out) hipLaunchKernelGGL((pre_srnet_kernel<float, true>), grid, block, 0, s, flow, est_prev, lr_t, (float*)out, hr_warp_out, B, H, W);
        else hipLaunchKernelGGL((pre_srnet_kernel<float, false>), grid, block, 0, s, flow, est_prev, lr_t, (float*)out, hr_warp_out, B, H, W);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ generic dense_image_warp (any C) and the 3-channel LR warp
__global__ void warp_generic_kernel(const float* __restrict__ img, const float* __restrict__ flow, float* __restrict__ out,
                                    int B, int H, int W, int C) {
    const long npix = (long)B * H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int x = p % W;
    long q = p / W;
    const int y = q % H;
    const int b = q / H;
    const float2 f = *reinterpret_cast<const float2*>(flow + p * 2);
    const WarpCoord cy = warp_coord((float)y - f.x, H);
    const WarpCoord cx = warp_coord((float)x - f.y, W);
    const float* r0 = img + (((size_t)b * H + cy.f) * W + cx.f) * C;
    const float* r1 = r0 + (size_t)W * C;
    for (int c = 0; c < C; ++c)
        out[p * C + c] = warp_interp(r0[c], r0[C + c], r1[c], r1[C + c], cx.a, cy.a);
}

hipError_t launch_warp_generic(const float* img, const float* flow, float* out, int B, int H, int W, int C, hipStream_t s) {
    const long npix = (long)B * H * W;
    hipLaunchKernelGGL(warp_generic_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, s, img, flow, out, B, H, W, C);
    return hipGetLastError();
}

hipError_t launch_warp3(const float* img, const float* flow, float* out, int B, int H, int W, hipStream_t s) {
    return launch_warp_generic(img, flow, out, B, H, W, 3, s);
}

// ------------------------------------------------------------------ TecoGAN epilogue: est = tanh(est + up4(lr))
__global__ void teco_epilogue_kernel(float* __restrict__ est, const float* __restrict__ lr, int B, int H, int W) {
    const int HH = 4 * H, WW = 4 * W;
    const long npix = (long)B * HH * WW;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int X = p % WW;
    long q = p / WW;
    const int Y = q % HH;
    const int b = q / HH;
    const int y0 = Y >> 2, x0 = X >> 2;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = (Y & 3) * 0.25f, tx = (X & 3) * 0.25f;
    const float* lb = lr + (size_t)b * H * W * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float up = lerp2(lb[((size_t)y0 * W + x0) * 3 + c], lb[((size_t)y0 * W + x1) * 3 + c],
                               lb[((size_t)y1 * W + x0) * 3 + c], lb[((size_t)y1 * W + x1) * 3 + c], tx, ty);
        est[p * 3 + c] = tanhf(est[p * 3 + c] + up);
    }
}

hipError_t launch_teco_epilogue(float* est, const float* lr_t, int B, int H, int W, hipStream_t s) {
    const long npix = (long)B * 16 * H * W;
    hipLaunchKernelGGL(teco_epilogue_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, s, est, lr_t, B, H, W);
    return hipGetLastError();
}

// ------------------------------------------------------------------ generic single ops (per-op entry points)
template <typename T>
__global__ void pad_channels_kernel(const float* __restrict__ in, T* __restrict__ out, long npix, int C, int Cp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * Cp) return;
    const long p = idx / Cp;
    const int c = idx % Cp;
    out[idx] = (T)(c < C ? in[p * C + c] : 0.f);
}

hipError_t launch_pad_channels(const float* in, void* out, int out_bf16, long npix, int C, int Cp, hipStream_t s) {
    const long total = npix * Cp;
    if (out_bf16) hipLaunchKernelGGL(pad_channels_kernel<__bf16>, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, (__bf16*)out, npix, C, Cp);
    else hipLaunchKernelGGL(pad_channels_kernel<float>, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, (float*)out, npix, C, Cp);
    return hipGetLastError();
}

__global__ void maxpool2_generic_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)B * Ho * Wo * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    long p = idx / C;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    float m = -INFINITY;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int y = 2 * oy + dy, x = 2 * ox + dx;
            if (y < H && x < W) m = fmaxf(m, in[(((size_t)b * H + y) * W + x) * C + c]);
        }
    out[idx] = m;
}

hipError_t launch_maxpool2_generic(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
    const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(maxpool2_generic_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, H, W, C);
    return hipGetLastError();
}

__global__ void resize_generic_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int sc) {
    const int Ho = sc * H, Wo = sc * W;
    const long total = (long)B * Ho * Wo * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    long p = idx / C;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int b = p / Ho;
    const float sy = (float)oy / (float)sc, sx = (float)ox / (float)sc;   // src = dst / s
    const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = sy - (float)y0, tx = sx - (float)x0;
    const float* base = in + (size_t)b * H * W * C + c;
    out[idx] = lerp2(base[((size_t)y0 * W + x0) * C], base[((size_t)y0 * W + x1) * C],
                     base[((size_t)y1 * W + x0) * C], base[((size_t)y1 * W + x1) * C], tx, ty);
}

hipError_t launch_resize_generic(const float* in, float* out, int B, int H, int W, int C, int sc, hipStream_t s) {
    const long total = (long)B * sc * H * sc * W * C;
    hipLaunchKernelGGL(resize_generic_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, H, W, C, sc);
    return hipGetLastError();
}

__global__ void s2d_generic_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int bs) {
    // out[b,h,w,(dy*bs+dx)*C + c] = in[b, bs*h+dy, bs*w+dx, c]
    const long total = (long)B * H * W * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    long p = idx / C;
    const int x = p % W;
    p /= W;
    const int y = p % H;
    const int b = p / H;
    const int Ho = H / bs, Wo = W / bs;
    const int oy = y / bs, dy = y % bs, ox = x / bs, dx = x % bs;
    out[(((size_t)b * Ho + oy) * Wo + ox) * (size_t)(bs * bs * C) + (dy * bs + dx) * C + c] = in[idx];
}

hipError_t launch_s2d_generic(const float* in, float* out, int B, int H, int W, int C, int bs, hipStream_t s) {
    const long total = (long)B * H * W * C;
    hipLaunchKernelGGL(s2d_generic_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, H, W, C, bs);
    return hipGetLastError();
}

// ------------------------------------------------------------------ LR synthesis: GaussianBlur 5x5 + cubic x1/4
// The reference makes its LR inputs with cv2.resize(cv2.GaussianBlur(hr, (5, 5), 0), ..., INTER_CUBIC)
// (FRVSR.py:117-119 in training, :274-277 in eval).  Restated (utils.blur_cubic_downscale, parity unpinned
// against OpenCV): blur taps g = [1,4,6,4,1]/16 with BORDER_REFLECT_101, then an exact x1/4 cubic reduction
// whose sample point 4d + 1.5 gives the weights k = [-3,19,19,-3]/32 on the four pixels of a block.  Both
// are separable, so per axis the pair is ONE 8-tap filter e[m] = sum_j k[j] g[m - j] on HR index 4d + m - 2,
// reflected per tap: 8 x 8 HR pixels per LR pixel in one pass instead of three numpy passes over the frames
// (236 ms per training batch on the host).
__global__ void blur_cubic_down4_kernel(const float* __restrict__ hr, float* __restrict__ lr, int N, int H, int W, int C) {
    const int h = H >> 2, w = W >> 2;
    const long total = (long)N * h * w * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    long p = idx / C;
    const int x = p % w;
    p /= w;
    const int y = p % h;
    const int n = p / h;
    // e = k (*) g = [-3, 7, 77, 175, 175, 77, 7, -3] / 512
    const float e[8] = {-3.f / 512, 7.f / 512, 77.f / 512, 175.f / 512, 175.f / 512, 77.f / 512, 7.f / 512, -3.f / 512};
    const float* img = hr + (size_t)n * H * W * C + c;
    int cols[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        int q = 4 * x + m - 2;
        q = q < 0 ? -q : (q >= W ? 2 * W - 2 - q : q);
        cols[m] = q * C;
    }
    float acc = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        int r = 4 * y + m - 2;
        r = r < 0 ? -r : (r >= H ? 2 * H - 2 - r : r);
        const float* row = img + (size_t)r * W * C;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += e[j] * row[cols[j]];
        acc += e[m] * t;
    }
    lr[idx] = acc;
}

__global__ void u8_normalize_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const uchar4 v = reinterpret_cast<const uchar4*>(in)[i];
    // IEEE division like numpy's astype(float32) / 255.0 (hipcc keeps fp32 division correctly rounded)
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x / 255.0f, v.y / 255.0f, v.z / 255.0f, v.w / 255.0f);
}

hipError_t launch_u8_normalize(const unsigned char* in, float* out, size_t n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(u8_normalize_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, in, out, n4);
    return hipGetLastError();
}

// ------------------------------------------------------------------ pad mode + uint8 output
__global__ void pad_copy_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int rows, int cols, int ch,
                                int rows_p, int cols_p) {
    const long total = (long)B * rows_p * cols_p * ch;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx % (cols_p * ch);
    long q = idx / (cols_p * ch);
    const int y = q % rows_p;
    const int b = q / rows_p;
    out[idx] = (y < rows && e < cols * ch) ? in[((size_t)b * rows + y) * cols * ch + e] : 0.f;
}

__global__ void crop_copy_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int rows, int cols, int ch,
                                 int rows_p, int cols_p) {
    const long total = (long)B * rows * cols * ch;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx % (cols * ch);
    long q = idx / (cols * ch);
    const int y = q % rows;
    const int b = q / rows;
    out[idx] = in[((size_t)b * rows_p + y) * cols_p * ch + e];
}

hipError_t launch_pad_copy(const float* in, float* out, int B, int rows, int cols, int ch, int rows_p, int cols_p, hipStream_t s) {
    const long total = (long)B * rows_p * cols_p * ch;
    hipLaunchKernelGGL(pad_copy_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, rows, cols, ch, rows_p, cols_p);
    return hipGetLastError();
}

hipError_t launch_crop_copy(const float* in, float* out, int B, int rows, int cols, int ch, int rows_p, int cols_p, hipStream_t s) {
    const long total = (long)B * rows * cols * ch;
    hipLaunchKernelGGL(crop_copy_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, rows, cols, ch, rows_p, cols_p);
    return hipGetLastError();
}

// 4 floats -> 4 bytes per thread; both row lengths (4w * 3 elements, padded and cropped) are multiples of 4
__global__ void denorm_u8_kernel(const float* __restrict__ est, unsigned char* __restrict__ out, int B, int HH, int row4_in,
                                 int rows, int row4_out) {
    const long total = (long)B * rows * row4_out;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx % row4_out;
    long q = idx / row4_out;
    const int y = q % rows;
    const int b = q / rows;
    const float4 v = reinterpret_cast<const float4*>(est)[((size_t)b * HH + y) * row4_in + e];
    // numpy's float32 -> uint8 cast truncates toward zero and wraps modulo 256
    uchar4 o;
    o.x = (unsigned char)(int)(v.x * 255.0f);
    o.y = (unsigned char)(int)(v.y * 255.0f);
    o.z = (unsigned char)(int)(v.z * 255.0f);
    o.w = (unsigned char)(int)(v.w * 255.0f);
    reinterpret_cast<uchar4*>(out)[idx] = o;
}

hipError_t launch_denorm_u8(const float* est, unsigned char* out, int B, int HH, int row_elems, int rows, int row_elems_out,
                            hipStream_t s) {
    if (row_elems % 4 || row_elems_out % 4 || row_elems_out > row_elems) return hipErrorInvalidValue;
    const long total = (long)B * rows * (row_elems_out / 4);
    hipLaunchKernelGGL(denorm_u8_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, est, out, B, HH, row_elems / 4, rows,
                       row_elems_out / 4);
    return hipGetLastError();
}

// cv2.resize(lr, 4x, INTER_CUBIC) (reference TecoGAN.py:313, FRVSR.py:304): destination pixel d samples the source at
// (d + 0.5) / 4 - 0.5, i.e. phase d & 3 has the fixed fraction {0.625, 0.875, 0.125, 0.375} and first tap (d >> 2) - 2 + (phase >> 1);
// 4-tap cubic kernel, a = -0.75, the last weight the remainder to 1, taps clamped to the border (utils.cubic_upscale4, parity
// unpinned against cv2).  Rows first, then columns, products and sums in the numpy restatement's order and without fused
// multiply-adds, so the two agree to the last bit.  wt: [4 phases][4 taps].
struct CubicW {
    float w[4][4];
};
__global__ void cubic_up4_kernel(const float* __restrict__ lr, float* __restrict__ out, int N, int h, int w, int C, CubicW cw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * 4 * h * 4 * w * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long p = idx / C;
    const int ox = (int)(p % (4 * w));
    p /= 4 * w;
    const int oy = (int)(p % (4 * h));
    const int n = (int)(p / (4 * h));
    const int py = oy & 3, px = ox & 3;
    const int y0 = (oy >> 2) - 2 + (py >> 1), x0 = (ox >> 2) - 2 + (px >> 1);
    const float* img = lr + (size_t)n * h * w * C + c;
    float col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xs = min(max(x0 + j, 0), w - 1);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ys = min(max(y0 + i, 0), h - 1);
            acc = __fadd_rn(acc, __fmul_rn(img[((size_t)ys * w + xs) * C], cw.w[py][i]));
        }
        col[j] = acc;
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __fadd_rn(acc, __fmul_rn(col[j], cw.w[px][j]));
    out[idx] = acc;
}

hipError_t launch_cubic_up4(const float* lr, float* out, int N, int h, int w, int C, hipStream_t s) {
    CubicW cw;
    const double a = -0.75;
    const double frac[4] = {0.625, 0.875, 0.125, 0.375};
    for (int ph = 0; ph < 4; ++ph) {
        const double t = frac[ph];
        const double w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a;
        const double w1 = ((a + 2) * t - (a + 3)) * t * t + 1;
        const double w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1;
        cw.w[ph][0] = (float)w0;
        cw.w[ph][1] = (float)w1;
        cw.w[ph][2] = (float)w2;
        cw.w[ph][3] = (float)(1.0 - (w0 + w1 + w2));
    }
    const long total = (long)N * 16 * h * w * C;
    hipLaunchKernelGGL(cubic_up4_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, lr, out, N, h, w, C, cw);
    return hipGetLastError();
}

hipError_t launch_blur_cubic_down4(const float* hr, float* lr, int N, int H, int W, int C, hipStream_t s) {
    const long total = (long)N * (H / 4) * (W / 4) * C;
    hipLaunchKernelGGL(blur_cubic_down4_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, hr, lr, N, H, W, C);
    return hipGetLastError();
}

}  // namespace sr
