// Backward kernels of the FRVSR training step (reference FRVSR.py:69-71,156: train_on_batch with two
// MSE losses and Adam), gfx950.  All tensors are fp32 NHWC with explicit channel strides.
//
//   wgrad         kernel + bias gradient of Conv2D 3x3 / Conv2DTranspose 3x3 s2: a GEMM whose K dimension
//                 is the PIXELS, so both operands are read straight from the NHWC tensors, 32 consecutive
//                 channels of one pixel per half wave (full 128-B lines, no staging, no transpose); one wave
//                 owns a 32 x 32 block of (cin, cout) for all 9 taps (144 accumulator registers).
//                 wgrad_x3_kernel (ctx bf16x3): v_mfma_f32_32x32x16_bf16, 16 pixels of a row per MFMA, operands
//                 split hi + lo once per row window; wgrad_kernel (ctx fp32): v_mfma_f32_32x32x2_f32; mode 2
//                 folds the taps of a <= 3-output conv into N.  Slices of the pixel range go to different
//                 waves, the four waves of a workgroup add up through LDS, a second kernel sums the
//                 partials in a fixed order (deterministic): per launch (wgrad_reduce_kernel) or, for a
//                 training step, all layers in one launch at the end (wgrad_reduce_batch_kernel).
//   dgrad_s2      data gradient of Conv2DTranspose s2 = 3x3 stride-2 convolution of dY with the same
//                 kernel; pixels are the M dimension, each lane streams contiguous channels of its pixel
//                 per tap; bf16x3 variant on forward-packed hi / lo weight fragments, fp32 variant on the
//                 Keras-layout tensor through L1; the producer's ReLU mask is folded into the store.
//   dgrad_few     data gradient of the convs with <= 3 outputs (HBM bound), activation mask in the store
//   (the data gradient of the other stride-1 convolutions runs on the forward MFMA kernel of conv_mfma.hip
//    with mirrored, transposed weights: sr_api.hip pack_entries PK_DGRAD; its MASK epilogue applies the
//    derivative of the activation whose output the gradient lands on)
//   act_bwd, maxpool2_bwd, resize_bwd, warp_bwd_flow, hr_warp_bwd, mse_grad: HBM-bound gathers; the pool and
//                 resize adjoints fold the LeakyReLU derivative of the tensor they write
//   adam, repack  optimizer update on the flat parameter vector and the re-pack of the updated
//                 kernels into the MFMA fragment order of the forward / dgrad convolutions (repack_batch:
//                 every buffer of the network in two launches, job table in the kernel arguments).
#include "kernels.h"
#include "gather_math.h"

#include <algorithm>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int blocks_for(long n, int bs) { return (int)((n + bs - 1) / bs); }

// ================================================================== wgrad
// modes: 0 Conv2D, 1 Conv2DTranspose s2, 2 Conv2D with <= 3 output channels (SRNet's last layer, FNet's
// flow head): the 9 taps are folded into the GEMM's N dimension (column j = tap*cout + co reads dY at the
// pixel shifted AGAINST the tap), one MFMA per k-step instead of nine with 29 of 32 columns empty.
enum { WG_CONV = 0, WG_CONVT = 1, WG_TAPSN = 2 };

struct WgradPlan {
    int mode;
    int R, S;            // rows / columns of one tap's matrix: conv (cin, cout), transposed conv (cout, cin)
    int nrb, ncolb, nblk, nslices;
    int part;            // floats of one (slice, block) partial: ntap x 32 x 32 accumulators + 64 bias lanes
    int x3;              // bf16x3 kernel: work is cut in segments of 16 pixels of one row
    int nsr;             // segments per row (x3)
};

static WgradPlan wgrad_plan(int B, int H, int W, int cin, int cout, int convt, int x3) {
    WgradPlan p;
    p.mode = convt ? WG_CONVT : (cout <= 3 ? WG_TAPSN : WG_CONV);
    p.x3 = x3 && p.mode != WG_TAPSN;
    p.nsr = (W + 15) / 16;
    p.R = convt ? cout : cin;
    p.S = convt ? cin : cout;
    p.nrb = (p.R + 31) / 32;
    p.ncolb = p.mode == WG_TAPSN ? 1 : (p.S + 31) / 32;
    p.nblk = p.nrb * p.ncolb;
    p.part = (p.mode == WG_TAPSN ? 1 : 9) * 1024 + 64;
    const long pairs = ((long)B * H * W + 1) / 2;
    // two waves per SIMD of 144 accumulator registers fill the chip with 2048 waves; the light taps-in-N
    // waves run eight deep
    long ns = (p.mode == WG_TAPSN ? 8192 : 2048) / p.nblk;
    if (ns > pairs / 16) ns = pairs / 16;         // at least 16 MFMA k-steps per slice
    if (p.x3) {
        const long nseg = (long)B * H * p.nsr;
        ns = 2048 / p.nblk;
        if (ns > nseg / 2) ns = nseg / 2;         // at least two 16-pixel segments per slice
    }
    if (ns < 1) ns = 1;
    // the four waves of a workgroup take four consecutive slices of ONE block and add their
    // accumulators through LDS before writing, so a launch leaves a quarter of the partials to reduce
    p.nslices = (int)((ns + 3) / 4);
    return p;
}

// workgroup -> (block, slice group): workgroup ids go round-robin over the 8 XCDs, so the
// XCD index is kept in the slice group and the blocks of one group (which read the same pixels) share an L2
struct WgId {
    int blk, grp;
};
__device__ __forceinline__ WgId wg_decode(const WgradPlan& pl) {
    const int q = blockIdx.x >> 3;
    WgId id;
    id.blk = q % pl.nblk;
    id.grp = (q / pl.nblk) * 8 + (blockIdx.x & 7);
    return id;
}
static int wg_grid(const WgradPlan& pl) { return 8 * pl.nblk * ((pl.nslices + 7) / 8); }

// sum of the four waves' accumulators in wave order (deterministic), result in wave 0
template <int NT>
__device__ __forceinline__ void wg_combine(f32x16 (&acc)[NT], float& bsum, int wave, int lane, float* red) {
    for (int k = 1; k < 4; ++k) {
        if (wave == k) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
            red[NT * 1024 + lane] = bsum;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
            bsum += red[NT * 1024 + lane];
        }
        __syncthreads();
    }
}

size_t wgrad_workspace_bytes(int B, int H, int W, int cin, int cout, int convt) {
    const WgradPlan p = wgrad_plan(B, H, W, cin, cout, convt, 0), q = wgrad_plan(B, H, W, cin, cout, convt, 1);
    return (size_t)std::max(p.nslices, q.nslices) * p.nblk * p.part * sizeof(float);
}

// Every load is unconditional from a clamped (always valid) address and masked afterwards: a branch per
// load would serialise them and force a full vmcnt(0) before the MFMAs; this way the next k-step's ten
// loads are in flight while the current one multiplies.
template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a, const WgradPlan pl) {
    constexpr int NT = MODE == WG_TAPSN ? 1 : 9;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kh = lane >> 5;
    __shared__ float red[NT * 1024 + 64];
    const WgId id = wg_decode(pl);
    const int blk = id.blk, slice = id.grp;
    if (slice >= pl.nslices) return;   // uniform for the workgroup
    const int wslice = slice * 4 + wave, nws = pl.nslices * 4;
    const int rb = blk / pl.ncolb, cb = blk - rb * pl.ncolb;
    const int H = a.H, W = a.W;
    const unsigned npix = (unsigned)a.B * H * W;
    const unsigned pairs = (npix + 1) / 2;
    const unsigned per = (pairs + nws - 1) / nws;
    const unsigned q0 = min((unsigned)wslice * per, pairs), q1 = min(q0 + per, pairs);
    // channel of this lane in the A (row) and B (column) operands
    const int ch_a = rb * 32 + i;
    const bool a_ok = ch_a < pl.R;
    int ch_b = cb * 32 + i;
    bool b_ok = ch_b < pl.S;
    int btap = 4;
    if constexpr (MODE == WG_TAPSN) {
        btap = min(i / pl.S, 8);
        ch_b = i - btap * pl.S;
        b_ok = i < 9 * pl.S;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    // this lane's pixel cursor p = 2q + kh as (row r = b*H + y, y, x); it advances by two pixels per k-step
    unsigned p = 2 * q0 + kh;
    int x = p % W;
    unsigned r = p / W;
    int y = r % H;
    const int srow = MODE == WG_CONVT ? 2 * W * a.ycs : W * a.xcs, scol = MODE == WG_CONVT ? a.ycs : a.xcs;
    const float* pa = (MODE == WG_CONVT ? a.dy : a.x) + (a_ok ? ch_a : 0);
    const float* pb = (MODE == WG_CONVT ? a.x : a.dy) + (b_ok ? ch_b : 0);
    const int bcs = MODE == WG_CONVT ? a.xcs : a.ycs;
    // taps-in-N: column (tap, co) reads dY at pixel - (tap offset)
    const int bdy = 1 - btap / 3, bdx = 1 - btap % 3;
    auto fetch = [&](float (&av)[NT], float& bv) {
        const bool valid = p < npix;
        const size_t pc = valid ? p : 0;
        if constexpr (MODE == WG_TAPSN) {
            const bool in = valid && b_ok && (unsigned)(y + bdy) < (unsigned)H && (unsigned)(x + bdx) < (unsigned)W;
            const float v = pb[(ptrdiff_t)(pc * bcs) + (in ? (bdy * W + bdx) * bcs : 0)];
            bv = in ? v : 0.f;
            const float u = pa[pc * a.xcs];
            av[0] = (valid && a_ok) ? u : 0.f;
        } else {
            const float v = pb[pc * bcs];
            bv = (valid && b_ok) ? v : 0.f;
            const bool ok = valid && a_ok;
            if constexpr (MODE == WG_CONV) {
                const float* c = pa + pc * a.xcs;
                const bool my[3] = {y > 0, true, y < H - 1}, mx[3] = {x > 0, true, x < W - 1};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const bool in = ok && my[t / 3] && mx[t % 3];
                    const float u = c[in ? (t / 3 - 1) * srow + (t % 3 - 1) * scol : 0];
                    av[t] = in ? u : 0.f;
                }
            } else {
                // dY pixel (2y + ky, 2x + kx) of image b: rows of all images are contiguous, 2r = b*2H + 2y
                const float* c = pa + (valid ? ((size_t)r * 4 * W + 2 * x) * a.ycs : 0);
                const bool my[3] = {true, true, y < H - 1}, mx[3] = {true, true, x < W - 1};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const bool in = ok && my[t / 3] && mx[t % 3];
                    const float u = c[in ? (t / 3) * srow + (t % 3) * scol : 0];
                    av[t] = in ? u : 0.f;
                }
            }
        }
    };
    auto advance = [&]() {
        p += 2;
        x += 2;
        while (x >= W) {
            x -= W;
            ++r;
            if (++y == H) y = 0;
        }
    };
    // k-steps go in groups of U: the next group's loads are in flight during this group's MFMAs.  The light
    // taps-in-N waves (one MFMA per k-step) are bound by load latency, so they keep four k-steps ahead.
    constexpr int U = MODE == WG_TAPSN ? 4 : 1;
    float av[U][NT], bv[U];
    auto load_group = [&](float (&A)[U][NT], float (&Bv)[U], unsigned q) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fetch(A[u], Bv[u]);
            if (q + u >= q1) {   // past this wave's slice (the pixels belong to the next one, or lie past the end)
                Bv[u] = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) A[u][t] = 0.f;
            }
            advance();
        }
    };
    load_group(av, bv, q0);
    for (unsigned q = q0; q < q1; q += U) {
        float an[U][NT], bn[U];
        load_group(an, bn, q + U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // bias: conv sums dY (the centre-tap columns in taps-in-N); transposed conv: the four phases
            // (taps 0, 1, 3, 4) cover every output pixel once
            if constexpr (MODE == WG_CONVT) bsum += av[u][0] + av[u][1] + av[u][3] + av[u][4];
            else bsum += bv[u];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) av[u][t] = an[u][t];
            bv[u] = bn[u];
        }
    }
    wg_combine<NT>(acc, bsum, wave, lane, red);
    if (wave != 0) return;
    float* out = a.part + ((size_t)slice * pl.nblk + blk) * pl.part;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = 8 * (rr >> 2) + 4 * kh + (rr & 3);
            out[t * 1024 + row * 32 + i] = acc[t][rr];
        }
    out[NT * 1024 + lane] = bsum;
}


// ---- bf16x3 variant (ctx dtype bf16x3): v_mfma_f32_32x32x16_bf16 takes 16 PIXELS per instruction, a lane
// (channel i, half kh) holding its channel at 8 consecutive pixels.  A wave works on segments of 16 pixels
// of one image row: each lane reads its channel at the pixels of its half plus the halo the taps need
// (conv: 10 pixels of rows y-1, y, y+1 of X; transposed conv: 18 pixels of rows 2y, 2y+1, 2y+2 of dY) --
// still one 128-B line per half wave and load -- splits every value ONCE into hi + lo bf16 and builds the
// three column taps of a row from the same registers (conv: windows shifted by 0, 1, 2 pixels; transposed
// conv: even, odd, even+2 pixels).  Per tap hi*hi + hi*lo + lo*hi, fp32 accumulate: 27 MFMAs of 32 cycles
// per 16 pixels instead of 72 fp32 MFMAs of 64 cycles.  Same partial layout and reduction as the fp32 kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_x3_kernel(const WgradArgs a, const WgradPlan pl) {
    constexpr int NW = MODE == WG_CONVT ? 18 : 10;   // window pixels per row per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kh = lane >> 5;
    __shared__ float red[9 * 1024 + 64];
    const WgId id = wg_decode(pl);
    const int blk = id.blk, slice = id.grp;
    if (slice >= pl.nslices) return;   // uniform for the workgroup
    const int rb = blk / pl.ncolb, cb = blk - rb * pl.ncolb;
    const int H = a.H, W = a.W;
    const unsigned nrows = (unsigned)a.B * H;
    const unsigned nseg = nrows * pl.nsr;
    const unsigned per = (nseg + pl.nslices * 4 - 1) / (pl.nslices * 4);
    const unsigned s0 = min((unsigned)(slice * 4 + wave) * per, nseg), s1 = min(s0 + per, nseg);
    const int ch_a = rb * 32 + i, ch_b = cb * 32 + i;
    const bool a_ok = ch_a < pl.R, b_ok = ch_b < pl.S;
    // shifted operand (A, rows of the tap matrix) and plain operand (B, columns)
    const float* pa = (MODE == WG_CONVT ? a.dy : a.x) + (a_ok ? ch_a : 0);
    const float* pb = (MODE == WG_CONVT ? a.x : a.dy) + (b_ok ? ch_b : 0);
    const int acs = MODE == WG_CONVT ? a.ycs : a.xcs, bcs = MODE == WG_CONVT ? a.xcs : a.ycs;
    const int AW = MODE == WG_CONVT ? 2 * W : W, AH = MODE == WG_CONVT ? 2 * H : H;   // dims of the shifted tensor
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    for (unsigned s = s0; s < s1; ++s) {
        const unsigned r = s / pl.nsr;                    // row b*H + y of the un-shifted tensor
        const int x0 = (s - r * pl.nsr) * 16 + 8 * kh;    // first pixel of this lane's half
        const int y = r % H;
        // plain operand: 8 pixels
        float u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = b_ok && x0 + e < W;
            const float v = pb[in ? ((size_t)r * W + x0 + e) * bcs : 0];
            u[e] = in ? v : 0.f;
        }
        // shifted operand: 3 rows x NW pixels
        float v[3][NW];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int yy = MODE == WG_CONVT ? 2 * y + rr : y - 1 + rr;              // row inside its image
            const size_t arow = MODE == WG_CONVT ? (size_t)2 * r + rr : (size_t)r + rr - 1;   // rows of all images are contiguous
            const bool rok = a_ok && yy >= 0 && yy < AH;
            const int xw0 = MODE == WG_CONVT ? 2 * x0 : x0 - 1;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const bool in = rok && xw0 + j >= 0 && xw0 + j < AW;
                const float t = pa[in ? (arow * AW + xw0 + j) * acs : 0];
                v[rr][j] = in ? t : 0.f;
            }
        }
        bf16x8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bh[e] = (__bf16)u[e];
            bl[e] = (__bf16)(u[e] - (float)bh[e]);
        }
        if constexpr (MODE == WG_CONVT) {
            // the four output phases (rows 2y, 2y+1; all 16 columns of the half) cover every dY pixel once
#pragma unroll
            for (int j = 0; j < 16; ++j) bsum += v[0][j] + v[1][j];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum += u[e];
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            __bf16 h[NW], l[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                h[j] = (__bf16)v[rr][j];
                l[j] = (__bf16)(v[rr][j] - (float)h[j]);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8 ah, al;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = MODE == WG_CONVT ? kx + 2 * e : kx + e;
                    ah[e] = h[j];
                    al[e] = l[j];
                }
                f32x16& c = acc[rr * 3 + kx];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            }
        }
    }
    wg_combine<9>(acc, bsum, wave, lane, red);
    if (wave != 0) return;
    float* out = a.part + ((size_t)slice * pl.nblk + blk) * pl.part;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = 8 * (rr >> 2) + 4 * kh + (rr & 3);
            out[t * 1024 + row * 32 + i] = acc[t][rr];
        }
    out[9 * 1024 + lane] = bsum;
}

// dw / db = sum of the slices' partials in slice order (deterministic).  One workgroup: 32 consecutive
// outputs x 8 slice lanes; lane g adds slices g, g + 8, ... and the eight sums meet in LDS in lane order.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw,
                                                  float* __restrict__ db, const WgradPlan& pl, int nbias, int block,
                                                  float (&red)[8][32]) {
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = block * 32 + e;
    const int nw = 9 * pl.R * pl.S;
    const size_t stride = (size_t)pl.nblk * pl.part;
    const int nacc = pl.part - 64;
    float sum = 0.f;
    if (idx < nw) {
        const int tap = idx / (pl.R * pl.S);
        const int rem = idx - tap * pl.R * pl.S;
        const int r = rem / pl.S, s = rem - r * pl.S;
        size_t off;
        if (pl.mode == WG_TAPSN) off = (size_t)(r >> 5) * pl.part + (r & 31) * 32 + tap * pl.S + s;
        else off = (size_t)((r >> 5) * pl.ncolb + (s >> 5)) * pl.part + tap * 1024 + (r & 31) * 32 + (s & 31);
        const float* p = part + off;
        // eight loads in flight per lane (the adds keep slice order): one at a time the kernel is latency-bound
        int sl = g;
        for (; sl + 56 < pl.nslices; sl += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sl + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; sl < pl.nslices; sl += 8) sum += p[sl * stride];
    } else if (idx < nw + nbias) {
        const int c = idx - nw;
        size_t off;
        if (pl.mode == WG_TAPSN) off = nacc + 4 * pl.S + c;                          // block 0, centre-tap columns
        else if (pl.mode == WG_CONVT) off = (size_t)((c >> 5) * pl.ncolb) * pl.part + nacc + (c & 31);
        else off = (size_t)(c >> 5) * pl.part + nacc + (c & 31);
        const float* p = part + off;
        for (int sl = g; sl < pl.nslices; sl += 8) sum += p[sl * stride] + p[sl * stride + 32];
    }
    red[g][e] = sum;
    __syncthreads();
    if (g == 0) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][e];
        if (idx < nw) dw[idx] = t;
        else if (idx < nw + nbias && db) db[idx - nw] = t;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           float* __restrict__ db, const WgradPlan pl, int nbias) {
    __shared__ float red[8][32];
    wgrad_reduce_body(part, dw, db, pl, nbias, blockIdx.x, red);
}

// the reductions of many layers in one launch (a training step defers all 38 of them to the end of the
// backward pass: each alone is a 10 us launch with too few workgroups to fill the chip)
struct WgradReduceJob {
    const float* part;
    float* dw;
    float* db;
    WgradPlan pl;
    int nbias, block0;
};
struct WgradReduceBatch {
    static constexpr int MAXJ = 48;   // 48 x 72 B of kernel arguments
    WgradReduceJob j[MAXJ];
    int njobs;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceBatch b) {
    __shared__ float red[8][32];
    int k = 0;
    for (int q = 1; q < b.njobs; ++q)
        if (b.j[q].block0 <= (int)blockIdx.x) k = q;
    const WgradReduceJob J = b.j[k];
    wgrad_reduce_body(J.part, J.dw, J.db, J.pl, J.nbias, (int)blockIdx.x - J.block0, red);
}

hipError_t launch_wgrad_reduce_batch(const WgradArgs* list, int n, hipStream_t s) {
    for (int first = 0; first < n; first += WgradReduceBatch::MAXJ) {
        WgradReduceBatch b;
        b.njobs = std::min(n - first, (int)WgradReduceBatch::MAXJ);
        int blocks = 0;
        for (int q = 0; q < b.njobs; ++q) {
            const WgradArgs& a = list[first + q];
            b.j[q] = {a.part, a.dw, a.db, wgrad_plan(a.B, a.H, a.W, a.cin, a.cout, a.convt, a.x3), a.cout, blocks};
            blocks += blocks_for(9 * a.cin * a.cout + a.cout, 32);
        }
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, s, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t s) {
    const WgradPlan pl = wgrad_plan(a.B, a.H, a.W, a.cin, a.cout, a.convt, a.x3);
    const dim3 grid(wg_grid(pl)), block(256);
    if (pl.x3 && pl.mode == WG_CONVT)
        hipLaunchKernelGGL(wgrad_x3_kernel<WG_CONVT>, grid, block, 0, s, a, pl);
    else if (pl.x3)
        hipLaunchKernelGGL(wgrad_x3_kernel<WG_CONV>, grid, block, 0, s, a, pl);
    else if (pl.mode == WG_CONVT)
        hipLaunchKernelGGL(wgrad_kernel<WG_CONVT>, grid, block, 0, s, a, pl);
    else if (pl.mode == WG_TAPSN)
        hipLaunchKernelGGL(wgrad_kernel<WG_TAPSN>, grid, block, 0, s, a, pl);
    else
        hipLaunchKernelGGL(wgrad_kernel<WG_CONV>, grid, block, 0, s, a, pl);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.defer_reduce) return e;
    const int n = 9 * a.cin * a.cout + a.cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks_for(n, 32)), dim3(256), 0, s, a.part, a.dw, a.db, pl, a.cout);
    return hipGetLastError();
}

// ================================================================== dgrad of Conv2DTranspose s2
// dX[y,x,ci] = sum_{ky,kx,co} dY[2y+ky, 2x+kx, co] * w[ky,kx,co,ci]     (dY = 0 outside [0,2H) x [0,2W))
// K = cout in blocks of 64 (lane (pixel i, half kh) holds channels 32*kh .. 32*kh+31 of its pixel,
// k-step s multiplies channel 32*kh + s), N = cin in blocks of NB*32 (blockIdx.y).
template <int NB>
__global__ __launch_bounds__(256) void dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, const float* __restrict__ relu_y,
                                                       int B, int H, int W, int K, int N, int ycs, int xcs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const unsigned npix = (unsigned)B * H * W;
    const unsigned p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= npix) return;
    const int n0 = blockIdx.y * NB * 32;
    const unsigned p = p0 + i;
    const bool valid = p < npix;
    const int x = p % W;
    const unsigned t1 = p / W;
    const int y = t1 % H;
    const int b = t1 / H;
    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // loads are unconditional from clamped addresses and masked afterwards (no branch per load), so the
    // compiler keeps the next tap's activations and the weight stream in flight under the MFMAs
    const size_t img = valid ? (size_t)b * 2 * H : 0;
    for (int t = 0; t < 9; ++t) {
        const int yy = 2 * y + t / 3, xx = 2 * x + t % 3;
        const bool in = valid && yy < 2 * H && xx < 2 * W;
        const float* src = dy + (in ? ((img + yy) * 2 * W + xx) * ycs : 0) + kh * 32;
        for (int kc = 0; kc < K; kc += 64) {
            f32x4 av[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 u = *reinterpret_cast<const f32x4*>(src + kc + 4 * v);
                av[v] = in ? u : z;
            }
            const float* wr = w + ((size_t)t * K + kc + kh * 32) * N + n0 + i;   // N % 32 == 0: every column exists
#pragma unroll
            for (int s = 0; s < 32; ++s) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const float bv = wr[(size_t)s * N + n * 32];
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s >> 2][s & 3], bv, acc[n], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = p0 + 8 * (r >> 2) + 4 * kh + (r & 3);
            const int ch = n0 + n * 32 + i;
            if (row < npix && ch < N) {
                float v = acc[n][r];
                // ReLU of the layer that produced this convolution's input, folded into the store
                if (relu_y && !(relu_y[(size_t)row * xcs + ch] > 0.f)) v = 0.f;
                dx[(size_t)row * xcs + ch] = v;
            }
        }
}

hipError_t launch_dgrad_s2(const float* dy, const float* w, float* dx, const float* relu_y, int B, int H, int W, int cin,
                           int cout, int ycs, int xcs, hipStream_t s) {
    if (cout % 64 || ycs % 4 || cin % 32) return hipErrorInvalidValue;
    const long npix = (long)B * H * W;
    const dim3 block(256);
    if (cin > 32) {
        if (cin % 64) return hipErrorInvalidValue;
        const dim3 grid(blocks_for(npix, 128), cin / 64);
        hipLaunchKernelGGL(dgrad_s2_kernel<2>, grid, block, 0, s, dy, w, dx, relu_y, B, H, W, cout, cin, ycs, xcs);
    } else {
        const dim3 grid(blocks_for(npix, 128), 1);
        hipLaunchKernelGGL(dgrad_s2_kernel<1>, grid, block, 0, s, dy, w, dx, relu_y, B, H, W, cout, cin, ycs, xcs);
    }
    return hipGetLastError();
}

// bf16x3 variant (ctx dtype bf16x3): v_mfma_f32_32x32x16_bf16, a lane (pixel i, half kh) holds 8 consecutive
// channels of its pixel (two 16-B loads, split hi + lo in registers); the weights are read as ready-made hi / lo
// B fragments from a buffer packed like a forward conv's (sr_api.hip pack_entries on the transposed conv's
// own kernel tensor, which read as HWIO IS the kernel of this stride-2 convolution: [chunk of 16 channels][tap]
// [n block][hi, lo][lane] x 16 B), kept current by the device re-pack.  6 MFMAs of 32 cycles per 16 channels
// and tap for 64 outputs instead of 16 fp32 MFMAs of 64 cycles.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NB>
__global__ __launch_bounds__(256) void dgrad_s2_x3_kernel(const float* __restrict__ dy, const u32x4* __restrict__ wpack,
                                                          float* __restrict__ dx, const float* __restrict__ relu_y,
                                                          int B, int H, int W, int K, int N, int ycs, int xcs, int wstride_cb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const unsigned npix = (unsigned)B * H * W;
    const unsigned p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= npix) return;
    const int n0 = blockIdx.y * NB * 32;
    const u32x4* wp = wpack + (size_t)blockIdx.y * wstride_cb + lane;
    const unsigned p = p0 + i;
    const bool valid = p < npix;
    const int x = p % W;
    const unsigned t1 = p / W;
    const int y = t1 % H;
    const int b = t1 / H;
    const int nchunk = K / 16;
    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const size_t img = valid ? (size_t)b * 2 * H : 0;
    for (int t = 0; t < 9; ++t) {
        const int yy = 2 * y + t / 3, xx = 2 * x + t % 3;
        const bool in = valid && yy < 2 * H && xx < 2 * W;
        const float* src = dy + (in ? ((img + yy) * 2 * W + xx) * ycs : 0) + kh * 8;
        for (int c = 0; c < nchunk; ++c) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(src + c * 16), u1 = *reinterpret_cast<const f32x4*>(src + c * 16 + 4);
            const f32x4 v0 = in ? u0 : z, v1 = in ? u1 : z;
            bf16x8 ah, al;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ah[e] = (__bf16)v0[e];
                al[e] = (__bf16)(v0[e] - (float)ah[e]);
                ah[4 + e] = (__bf16)v1[e];
                al[4 + e] = (__bf16)(v1[e] - (float)ah[4 + e]);
            }
            const u32x4* wv = wp + (size_t)(c * 9 + t) * NB * 128;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const bf16x8 wh = __builtin_bit_cast(bf16x8, wv[n * 128]), wl = __builtin_bit_cast(bf16x8, wv[n * 128 + 64]);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = p0 + 8 * (r >> 2) + 4 * kh + (r & 3);
            const int ch = n0 + n * 32 + i;
            if (row < npix && ch < N) {
                float v = acc[n][r];
                if (relu_y && !(relu_y[(size_t)row * xcs + ch] > 0.f)) v = 0.f;
                dx[(size_t)row * xcs + ch] = v;
            }
        }
}

// wpack: bf16x3 buffer of a forward conv planned as (cin = cout of the transposed conv, cout = its cin); nt its
// n blocks per output-channel block (1 or 2), wstride_cb its vectors per output-channel block
hipError_t launch_dgrad_s2_x3(const float* dy, const void* wpack, int nt, int wstride_cb, float* dx, const float* relu_y, int B,
                              int H, int W, int cin, int cout, int ycs, int xcs, hipStream_t s) {
    if (cout % 16 || ycs % 4 || cin % 32 || (nt != 1 && nt != 2)) return hipErrorInvalidValue;
    const long npix = (long)B * H * W;
    const dim3 block(256), grid(blocks_for(npix, 128), (cin + 32 * nt - 1) / (32 * nt));
    if (nt == 2)
        hipLaunchKernelGGL(dgrad_s2_x3_kernel<2>, grid, block, 0, s, dy, (const u32x4*)wpack, dx, relu_y, B, H, W, cout, cin, ycs, xcs, wstride_cb);
    else
        hipLaunchKernelGGL(dgrad_s2_x3_kernel<1>, grid, block, 0, s, dy, (const u32x4*)wpack, dx, relu_y, B, H, W, cout, cin, ycs, xcs, wstride_cb);
    return hipGetLastError();
}

// ================================================================== data gradient of a conv with <= 3 outputs
// dX[y,x,ci] = sum_{ky,kx,co} dY[y+1-ky, x+1-kx, co] * w[ky,kx,ci,co]: K = 9*cout <= 27 products per output
// element, so the layer (SRNet's 64->3 at HR resolution, FNet's 32->2) is bound by WRITING dX (bound: HBM):
// plain FMAs, the 27 x cin weights in LDS, the 27 gathered dY values in registers, 16 channels per thread,
// and the derivative of the activation that produced this conv's input folded into the store (act / y),
// which saves the separate mask pass over the largest tensor of the step (4h x 4w x 64).
// One workgroup = 64 consecutive pixels x (cin / 16) waves; a wave owns 16 output channels, so its weights are
// wave-uniform and come through scalar loads straight into the FMAs (no LDS), a lane owns one pixel and
// reads the dY taps as one 16-byte load each (the four waves share them in L1).
template <int COUT>   // compile-time: the wave's 16 x COUT weights of a tap are one contiguous run of scalar loads
__global__ __launch_bounds__(256) void dgrad_few_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                        float* __restrict__ dx, const float* __restrict__ y, int act,
                                                        int B, int H, int W, int cin, int gcs, int xcs) {
    // the workgroup's 64 pixels x cin channels, transposed on the way in (the activation output y) and out (dX): a lane owns 64
    // bytes of its pixel, so from registers one instruction would touch 16 bytes of each of 64 lines (measured: 1.44 GB written
    // for a 671 MB tensor); through LDS every instruction moves whole lines
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * 256];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = wv * 16;
    const long npix = (long)B * H * W;
    const long p0 = (long)blockIdx.x * 64;
    const long p = p0 + lane;
    const bool live = p < npix;
    const long pc = live ? p : npix - 1;
    const int x = pc % W;
    const int yy = (pc / W) % H;
    // LDS row = pixel (cin * 4 bytes), 16-byte pieces XOR-swizzled by the pixel so that the 64 lanes of an access spread over the banks
    const int row_bytes = cin * 4, pieces = cin >> 2;            // pieces per pixel: 4, 8, 12 or 16
    const int sm = pieces == 16 ? 15 : (pieces == 8 ? 7 : 3);    // the XOR stays inside the row
    // piece 64 * (cin / 16) * k + thread of the tile (k < 4): the tile is contiguous in memory when xcs == cin
    int prow[4], ppc[4];
    f32x4 mreg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int P = k * (int)blockDim.x + (int)threadIdx.x;
        prow[k] = P / pieces;
        ppc[k] = P - prow[k] * pieces;
        mreg[k] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (y && p0 + prow[k] < npix) mreg[k] = *reinterpret_cast<const f32x4*>(y + (p0 + prow[k]) * xcs + 4 * ppc[k]);   // lands under the FMAs
    }
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int sy = yy + 1 - tap / 3, sx = x + 1 - tap % 3;
        const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
        const f32x4 u = *reinterpret_cast<const f32x4*>(g + (in ? (pc + (1 - tap / 3) * W + (1 - tap % 3)) * gcs : 0));
        const float* wt = w + ((size_t)tap * cin + c0) * COUT;   // wave-uniform: scalar loads
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float gv = in ? u[co] : 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += gv * wt[c * COUT + co];
        }
    }
    const float neg = act == A_LRELU ? 0.2f : 0.f;
    if (y) {   // the activation output of the tile, read in full lines above, to the lanes that own it
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<f32x4*>(tile + prow[k] * row_bytes + ((ppc[k] ^ (prow[k] & sm)) << 4)) = mreg[k];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) mreg[q] = *reinterpret_cast<const f32x4*>(tile + lane * row_bytes + (((4 * wv + q) ^ (lane & sm)) << 4));
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        if (y) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mreg[q][e] > 0.f ? 1.f : neg;
        }
        *reinterpret_cast<f32x4*>(tile + lane * row_bytes + (((4 * wv + q) ^ (lane & sm)) << 4)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (p0 + prow[k] < npix)
            *reinterpret_cast<f32x4*>(dx + (p0 + prow[k]) * xcs + 4 * ppc[k]) =
                *reinterpret_cast<const f32x4*>(tile + prow[k] * row_bytes + ((ppc[k] ^ (prow[k] & sm)) << 4));
}

// cin 16 .. 64 in steps of 16, cout <= 3, gcs and xcs multiples of 4 (dY rows are read 16 bytes at a time, so the
// tensor must be padded to 4 channels); y (nullable, laid out like dx) with act A_RELU / A_LRELU
hipError_t launch_dgrad_few(const float* g, const float* w, float* dx, const float* y, int act, int B, int H, int W, int cin,
                            int cout, int gcs, int xcs, hipStream_t s) {
    if (cin % 16 || cin > 64 || cout > 3 || cout < 1 || xcs % 4 || gcs % 4 || (y && act != A_RELU && act != A_LRELU))
        return hipErrorInvalidValue;
    const long npix = (long)B * H * W;
    const dim3 grid(blocks_for(npix, 64)), block(64 * (cin / 16));
    if (cout == 3) hipLaunchKernelGGL(dgrad_few_kernel<3>, grid, block, 0, s, g, w, dx, y, act, B, H, W, cin, gcs, xcs);
    else if (cout == 2) hipLaunchKernelGGL(dgrad_few_kernel<2>, grid, block, 0, s, g, w, dx, y, act, B, H, W, cin, gcs, xcs);
    else hipLaunchKernelGGL(dgrad_few_kernel<1>, grid, block, 0, s, g, w, dx, y, act, B, H, W, cin, gcs, xcs);
    return hipGetLastError();
}

// ================================================================== activation derivative from the OUTPUT
// g[p, c] *= act'(.) expressed through y = act(.):  ReLU y > 0; LeakyReLU(0.2) y > 0 ? 1 : 0.2;
// tanh 1 - y^2; clip[0,1] 0 < y < 1  (Keras / TF and torch agree away from the kinks)
__global__ void act_bwd_kernel(float* __restrict__ g, const float* __restrict__ y, int act, long n, int C, int gcs, int ycs) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const long p = idx / C;
    const int c = idx - p * C;
    const float v = y[p * ycs + c];
    float d = 1.f;
    if (act == A_RELU) d = v > 0.f ? 1.f : 0.f;
    else if (act == A_LRELU) d = v > 0.f ? 1.f : 0.2f;
    else if (act == A_TANH) d = 1.f - v * v;
    else if (act == A_CLIP01) d = (v > 0.f && v < 1.f) ? 1.f : 0.f;
    g[p * gcs + c] *= d;
}

hipError_t launch_act_bwd(float* g, const float* y, int act, long npix, int C, int gcs, int ycs, hipStream_t s) {
    const long n = npix * C;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, g, y, act, n, C, gcs, ycs);
    return hipGetLastError();
}

// ================================================================== MaxPool2D 2x2 SAME backward
// the gradient goes to the first maximum of the window in row-major order (TF MaxPoolGrad, torch)
// act (A_RELU / A_LRELU, else none): x is that activation's output, its derivative is folded into the store
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                    int B, int H, int W, int C, int act) {
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
    float g = gy[idx];
    int best = -1;
    float bv = 0.f;
    for (int k = 0; k < 4; ++k) {
        const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
        if (yy >= H || xx >= W) continue;
        const float v = x[(((size_t)b * H + yy) * W + xx) * C + c];
        if (best < 0 || v > bv) {
            best = k;
            bv = v;
        }
    }
    if (act == A_RELU) g = bv > 0.f ? g : 0.f;
    else if (act == A_LRELU) g = bv > 0.f ? g : 0.2f * g;
    for (int k = 0; k < 4; ++k) {
        const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
        if (yy >= H || xx >= W) continue;
        gx[(((size_t)b * H + yy) * W + xx) * C + c] = k == best ? g : 0.f;
    }
}

hipError_t launch_maxpool2_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int act,
                               hipStream_t s) {
    const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, x, gy, gx, B, H, W, C, act);
    return hipGetLastError();
}

// ================================================================== adjoint of the TF1-legacy bilinear resize
// forward: out[Y] = in[lo] + (in[hi] - in[lo]) * t, lo = Y / s, hi = min(lo + 1, n - 1), t = (Y % s) / s.
// gather form: input row y collects weight (1 - t) from the s output rows of its own block, t from the
// block above (whose hi is y), and t again from its own block when y is the last row (hi clamps to y).
__device__ __forceinline__ float resize_adj_weight(int Y, int y, int n, int s) {
    const int lo = Y / s;
    const float t = (float)(Y - lo * s) / (float)s;
    float w = 0.f;
    if (lo == y) w += 1.f - t;
    if (min(lo + 1, n - 1) == y) w += t;
    return w;
}

// ym (or nullptr): output [B,H,W,C] of the ReLU / LeakyReLU `act` that fed the resize; its derivative scales the store
__global__ void resize_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B, int H, int W, int C,
                                  int s, int ocs, int accumulate, const float* __restrict__ ym, int act) {
    const long total = (long)B * H * W * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C;
    long p = idx / C;
    const int x = p % W;
    p /= W;
    const int y = p % H;
    const int b = p / H;
    const int Ho = s * H, Wo = s * W;
    const int Y0 = max(0, s * (y - 1)), Y1 = min(Ho, s * (y + 1));
    const int X0 = max(0, s * (x - 1)), X1 = min(Wo, s * (x + 1));
    float sum = 0.f;
    for (int Y = Y0; Y < Y1; ++Y) {
        const float wy = resize_adj_weight(Y, y, H, s);
        float row = 0.f;
        for (int X = X0; X < X1; ++X)
            row += resize_adj_weight(X, x, W, s) * gy[(((size_t)b * Ho + Y) * Wo + X) * C + c];
        sum += wy * row;
    }
    if (ym) sum = ym[idx] > 0.f ? sum : (act == A_LRELU ? 0.2f * sum : 0.f);
    float* o = gx + (((size_t)b * H + y) * W + x) * ocs + c;
    *o = accumulate ? *o + sum : sum;
}

hipError_t launch_resize_bwd(const float* gy, float* gx, int B, int H, int W, int C, int sc, int ocs, int accumulate,
                             const float* ym, int act, hipStream_t s) {
    if (ym && act != A_RELU && act != A_LRELU) return hipErrorInvalidValue;
    const long total = (long)B * H * W * C;
    hipLaunchKernelGGL(resize_bwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, gy, gx, B, H, W, C, sc, ocs, accumulate,
                       ym, act);
    return hipGetLastError();
}

// ================================================================== dense_image_warp, gradient w.r.t. the flow
// out = ay*(bot - top) + top, top = ax*(tr - tl) + tl, bot = ax*(br - bl) + bl, a = clamp(q - floor, 0, 1),
// q = coordinate - flow: d out / d flow_y = -(bot - top), d out / d flow_x = -((1-ay)(tr-tl) + ay(br-bl)),
// passed where the clamp is inactive (0 <= q - floor <= 1, inclusive like tf.clip_by_value / torch.clamp);
// the floor and its clamp have zero gradient, the image is a constant input (reference FRVSR.py:154).
struct WarpGrad {
    float gy, gx;
};
__device__ __forceinline__ void warp_grad_acc(WarpGrad& o, float g, float tl, float tr, float bl, float br, float ax, float ay) {
    const float top = ax * (tr - tl) + tl;
    const float bot = ax * (br - bl) + bl;
    o.gy -= g * (bot - top);
    o.gx -= g * ((1.f - ay) * (tr - tl) + ay * (br - bl));
}
__device__ __forceinline__ bool warp_pass(float q, int n) {
    float f = floorf(q);
    f = fminf(fmaxf(f, 0.f), (float)(n - 2));
    const float d = q - f;
    return d >= 0.f && d <= 1.f;
}

__global__ void warp_bwd_flow_kernel(const float* __restrict__ img, const float* __restrict__ flow, const float* __restrict__ g,
                                     float* __restrict__ gflow, int B, int H, int W, int C, int gfs, int accumulate) {
    const long npix = (long)B * H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int x = p % W;
    long q = p / W;
    const int y = q % H;
    const int b = q / H;
    const float2 f = *reinterpret_cast<const float2*>(flow + p * 2);
    const float qy = (float)y - f.x, qx = (float)x - f.y;
    const WarpCoord cy = warp_coord(qy, H);
    const WarpCoord cx = warp_coord(qx, W);
    const float* r0 = img + (((size_t)b * H + cy.f) * W + cx.f) * C;
    const float* r1 = r0 + (size_t)W * C;
    WarpGrad o = {0.f, 0.f};
    for (int c = 0; c < C; ++c) warp_grad_acc(o, g[p * C + c], r0[c], r0[C + c], r1[c], r1[C + c], cx.a, cy.a);
    if (!warp_pass(qy, H)) o.gy = 0.f;
    if (!warp_pass(qx, W)) o.gx = 0.f;
    float* out = gflow + p * gfs;
    out[0] = accumulate ? out[0] + o.gy : o.gy;
    out[1] = accumulate ? out[1] + o.gx : o.gx;
}

hipError_t launch_warp_bwd_flow(const float* img, const float* flow, const float* g, float* gflow, int B, int H, int W,
                                int C, int gfs, int accumulate, hipStream_t s) {
    const long npix = (long)B * H * W;
    hipLaunchKernelGGL(warp_bwd_flow_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, s, img, flow, g, gflow, B, H, W, C, gfs, accumulate);
    return hipGetLastError();
}

// HR warp of pre_srnet (fused_ops.hip): the gradient of SRNet's input channels 3 + (dy*4+dx)*3 + c
// (space_to_depth order, reference layers.py:109) w.r.t. F_HR at HR pixel (4y+dy, 4x+dx), F_HR being
// recomputed from the LR flow with the forward's arithmetic so that every floor falls the same way.
__global__ void hr_warp_bwd_kernel(const float* __restrict__ est, const float* __restrict__ flow, const float* __restrict__ g_sin,
                                   float* __restrict__ g_fhr, int B, int H, int W, int gcs) {
    const int HH = 4 * H, WW = 4 * W;
    const long npix = (long)B * HH * WW;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int X = p % WW;
    long q = p / WW;
    const int Y = q % HH;
    const int b = q / HH;
    const int y = Y >> 2, x = X >> 2, dy = Y & 3, dx = X & 3;
    const int x1 = min(x + 1, W - 1), y1 = min(y + 1, H - 1);
    const float* fb = flow + (size_t)b * H * W * 2;
    const float2 ftl = *reinterpret_cast<const float2*>(fb + ((size_t)y * W + x) * 2);
    const float2 ftr = *reinterpret_cast<const float2*>(fb + ((size_t)y * W + x1) * 2);
    const float2 fbl = *reinterpret_cast<const float2*>(fb + ((size_t)y1 * W + x) * 2);
    const float2 fbr = *reinterpret_cast<const float2*>(fb + ((size_t)y1 * W + x1) * 2);
    const float ty = dy * 0.25f, tx = dx * 0.25f;
    const float fy = lerp2(ftl.x, ftr.x, fbl.x, fbr.x, tx, ty);
    const float fx = lerp2(ftl.y, ftr.y, fbl.y, fbr.y, tx, ty);
    const float qy = (float)Y - fy, qx = (float)X - fx;
    const WarpCoord cy = warp_coord(qy, HH);
    const WarpCoord cx = warp_coord(qx, WW);
    const float* r0 = est + (((size_t)b * HH + cy.f) * WW + cx.f) * 3;
    const float* r1 = r0 + (size_t)WW * 3;
    const float* g = g_sin + (((size_t)b * H + y) * W + x) * gcs + 3 + (dy * 4 + dx) * 3;
    WarpGrad o = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) warp_grad_acc(o, g[c], r0[c], r0[3 + c], r1[c], r1[3 + c], cx.a, cy.a);
    if (!warp_pass(qy, HH)) o.gy = 0.f;
    if (!warp_pass(qx, WW)) o.gx = 0.f;
    *reinterpret_cast<float2*>(g_fhr + p * 2) = make_float2(o.gy, o.gx);
}

hipError_t launch_hr_warp_bwd(const float* est_prev, const float* flow, const float* g_sin, float* g_fhr, int B, int H, int W,
                              int gcs, hipStream_t s) {
    const long npix = (long)B * H * W * 16;
    hipLaunchKernelGGL(hr_warp_bwd_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, s, est_prev, flow, g_sin, g_fhr, B, H, W, gcs);
    return hipGetLastError();
}

// ================================================================== mean-squared-error loss and its gradient
// Keras: mean over every element (reference FRVSR.py:70).  g[p, c < C] = scale * (pred - target), times the
// derivative of the output non-linearity: tanh_out (TecoGAN's epilogue, TecoGAN.py:76): 1 - pred^2;
// clip_src (SRNet's final clip[0,1], reference layers.py:76; = pred for FRVSR, the tensor before the
// epilogue for TecoGAN): zero where it sits on a clip bound.  g[p, C..gcs) = 0;
// partial[block] = sum (pred - target)^2 of the block's elements.
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       float* __restrict__ g, double* __restrict__ partial, long npix, int C,
                                                       int gcs, float scale, const float* __restrict__ clip_src, int tanh_out) {
    __shared__ double red[256];
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double sq = 0.0;
    if (idx < npix * gcs) {
        const long p = idx / gcs;
        const int c = idx - p * gcs;
        float gv = 0.f;
        if (c < C) {
            const float a = pred[p * C + c];
            const float d = a - target[p * C + c];
            sq = (double)d * d;
            gv = scale * d;
            if (tanh_out) gv *= 1.f - a * a;
            if (clip_src) {
                const float u = clip_src[p * C + c];
                if (!(u > 0.f && u < 1.f)) gv = 0.f;
            }
        }
        g[idx] = gv;
    }
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double inv_count, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] * inv_count);
}

size_t mse_workspace_bytes(long npix, int gcs) { return (size_t)blocks_for(npix * gcs, 256) * sizeof(double); }

hipError_t launch_mse_grad(const float* pred, const float* target, float* g, double* partial, float* loss_out, long npix, int C,
                           int gcs, const float* clip_src, int tanh_out, hipStream_t s) {
    const int nb = blocks_for(npix * gcs, 256);
    const double count = (double)npix * C;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(nb), dim3(256), 0, s, pred, target, g, partial, npix, C, gcs, (float)(2.0 / count), clip_src, tanh_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, nb, 1.0 / count, loss_out);
    return hipGetLastError();
}

// ================================================================== Adam (Keras 2.2.4 optimizers.py) and re-pack
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                            long n, float lr_t, float b1, float b2, float eps, float gscale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

hipError_t launch_adam(float* p, float* m, float* v, const float* g, long n, float lr_t, float b1, float b2, float eps,
                       float gscale, hipStream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, p, m, v, g, n, lr_t, b1, b2, eps, gscale);
    return hipGetLastError();
}

// packed[dst] = kernel[src] for the listed elements (padding zeros never change); bf16 rounds to nearest
// even exactly like the host packer, bf16x3 also writes the residual 64 vectors further on
__device__ __forceinline__ unsigned short f2bf_dev(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void repack_kernel(const float* __restrict__ kernel, const int2* __restrict__ ents, int n, void* __restrict__ out, int prec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 e = ents[i];
    const float v = kernel[e.x];
    const unsigned dst = (unsigned)e.y;
    if (prec == P_FP32) {
        reinterpret_cast<float*>(out)[dst] = v;
    } else {
        unsigned short* o = reinterpret_cast<unsigned short*>(out);
        const unsigned short hi = f2bf_dev(v);
        o[dst] = hi;
        if (prec == P_BF16X3) o[dst + 512] = f2bf_dev(v - __uint_as_float((unsigned)hi << 16));
    }
}

hipError_t launch_repack(const float* kernel, const void* ents, int n, void* out, int prec, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(repack_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, kernel, (const int2*)ents, n, out, prec);
    return hipGetLastError();
}

// Many re-packs in one launch (after Adam every conv's forward, data-gradient and projection buffer plus its
// bias follow the master weights: ~110 jobs of a few thousand elements, which as separate launches cost more
// in launch latency than in work).  The job table travels in the kernel arguments; a workgroup finds its job
// by a uniform scan (scalar loads of the argument segment).
__global__ void repack_batch_kernel(const float* __restrict__ params, const RepackBatch b, int prec) {
    int k = 0;
    for (int q = 1; q < b.njobs; ++q)
        if (b.j[q].block0 <= (int)blockIdx.x) k = q;
    const RepackJob J = b.j[k];
    const int i = ((int)blockIdx.x - J.block0) * 256 + threadIdx.x;
    if (i >= J.n) return;
    if (!J.ents) {   // bias: plain copy
        reinterpret_cast<float*>(J.out)[i] = params[J.src + i];
        return;
    }
    const int2 e = reinterpret_cast<const int2*>(J.ents)[i];
    const float v = params[J.src + e.x];
    const unsigned dst = (unsigned)e.y;
    if (prec == P_FP32) {
        reinterpret_cast<float*>(J.out)[dst] = v;
    } else {
        unsigned short* o = reinterpret_cast<unsigned short*>(J.out);
        const unsigned short hi = f2bf_dev(v);
        o[dst] = hi;
        if (prec == P_BF16X3) o[dst + 512] = f2bf_dev(v - __uint_as_float((unsigned)hi << 16));
    }
}

hipError_t launch_repack_batch(const float* params, RepackBatch& b, int prec, hipStream_t s) {
    if (b.njobs <= 0) return hipSuccess;
    if (b.njobs > RepackBatch::MAXJ) return hipErrorInvalidValue;
    int blocks = 0;
    for (int q = 0; q < b.njobs; ++q) {
        if (b.j[q].n <= 0 || !b.j[q].out) return hipErrorInvalidValue;
        b.j[q].block0 = blocks;
        blocks += (b.j[q].n + 255) / 256;
    }
    hipLaunchKernelGGL(repack_batch_kernel, dim3(blocks), dim3(256), 0, s, params, b, prec);
    return hipGetLastError();
}

}  // namespace sr
