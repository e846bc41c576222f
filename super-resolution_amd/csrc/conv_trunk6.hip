// SRNet trunk, second arithmetic: the 20 residual-block convs (reference layers.py:58-62) with the fp32 product split as
//     w * x  ~=  f16(w) * f16(x)  +  fp6(w) * fp6(x - f16(x))  +  fp6(w - f16(w)) * fp6(x)
// one v_mfma_f32_32x32x16_f16 per 16 channels for the first term and one MX-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (FP6 E2M3,
// one E8M0 scale per 32 channels) per 64 channels for each cross term: 6 MFMA issues per tap and 64 channels where the
// bf16x3 kernel of conv_trunk.hip needs 12, at the same 32 cycles each (tools/probe/mx_rate_probe.hip).  The cross terms are
// 2^-11 of the product, so 4 significant bits on them leave the sum at ~2^-15 relative: measured against the fp64 oracle in
// tests/test_gpu_mx6.py (bar 1e-3 on [0, 1] pixels).
//
// Data layout ("mx6"): [B][4 planes][2 halves][H][W][32 B] -- per pixel the same four planes of four 16-byte units as a
// split-planar tensor, but units 2 h, 2 h + 1 of a plane (what lane half h reads, and what one lane writes of the FP6 planes) in a
// half plane of their own.  What one lane loads or stores in the epilogue is 16 bytes: with 64-byte pixel records an instruction
// touched 64 cache lines (one per lane) and the tag lookups of those 14 instructions per tile and wave cost as much as the MFMAs
// (DESIGN.md 4.1d); with 32-byte records it touches 16, and a tile DMA instruction still 8 (16 pixels x 32 contiguous bytes from
// each of two half planes; one plane per 16-byte unit makes the epilogue cheaper still and the DMA twice as slow).  The DMA
// gathers a pixel's four units into the 64-byte LDS record of conv_trunk.hip (the source address of a global_load_lds lane is
// free, the LDS side is linear), so the swizzle and every LDS read address carry over.  Per pixel:
//   plane P = 0, 1: f16 hi parts of channels 32 P .. 32 P + 31; unit u = 2 h + ks (16 bytes) holds k-step ks of lane half h:
//                   element e = channel 32 P + 16 ks + split_perm(h, e)
//   plane 2 ("x6"): FP6 of the value itself; bytes [32 h, 32 h + 24) = the 32 channels of block h, element e (6 bits at bit 6 e)
//                   = channel 32 h + c6(e), c6(e) = ((e >> 1) & 3) + 8 (e >> 3) + 4 (e & 1); byte 32 h + 24 = E8M0 scale, rest 0
//   plane 3 ("lo6"): FP6 of value - f16(value), same record
// so a lane of the B operand reads units 2 h and 2 h + 1 of its pixel in every plane step, exactly like the bf16x3 kernel.
// A stored activation is hi + lo6 (the residual stream is read back that way); 4 bytes per element like fp32.
//
// The odd element orders are what the epilogue produces without a transpose: after the last plane step the two lane halves swap
// accumulators (v_permlane32_swap) so that lane (r, h) holds all 32 channels of ONE pixel (M block h, column r) in accumulator
// order, v_cvt_scalef32_2xpk16_fp6_f32 interleaves the two halves' registers into one 24-byte FP6 block, and 8 consecutive
// registers are one f16 unit.
#include "kernels.h"
#include "mx6_format.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define SR_AS1 __attribute__((address_space(1)))
#define SR_AS3 __attribute__((address_space(3)))

using namespace mx6;

namespace {

struct T6 {
    static constexpr int TH = 8, TWX = 64;
    static constexpr int HR = TH + 2, HC = TWX + 2;
    static constexpr int NPIX = HR * HC;                    // 660
    static constexpr int NUNIT = NPIX * 4;
    static constexpr int NINST = (NUNIT + 63) / 64;         // 42 DMA wave-instructions (1 KiB each)
    static constexpr int PER_WAVE = (NINST + 7) / 8;        // 6 (waves 0, 1) or 5
    static constexpr int A_BYTES = NINST * 1024;            // 43,008 per buffer
    static constexpr int W_BYTES = 4 * 9 * 2 * 1024;        // 73,728: [step][tap][2 units][lane] x 16 B
    static constexpr int BIAS_OFF = W_BYTES + 2 * A_BYTES;  // 32 floats
    static constexpr int LDS = BIAS_OFF + 128;              // 159,872 <= 163,840
};

// the same 32 channels as two split-planar (conv_trunk.hip) plane records: what conv2d_transpose_1 reads
template <int PL>
__device__ __forceinline__ void encode_split_plane(const f32x16& V0, const f32x16& V1, u32x4 (&o)[4]) {      // plane PL of the two
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const f32x16& V = hh ? V1 : V0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = V[8 * PL + 2 * k], b = V[8 * PL + 2 * k + 1];
            const bf16x2 hv = {(__bf16)a, (__bf16)b};
            const unsigned hu = __builtin_bit_cast(unsigned, hv);
            const bf16x2 lv = {(__bf16)(a - __builtin_bit_cast(float, hu << 16)), (__bf16)(b - __builtin_bit_cast(float, hu & 0xFFFF0000u))};
            o[2 * hh][k] = hu;
            o[2 * hh + 1][k] = __builtin_bit_cast(unsigned, lv);
        }
    }
}

}  // namespace

// RELU: ReLU after the bias; RES: add a.res (mx6) after it.
//
// Same skeleton as trunk_conv_kernel: persistent 512-thread workgroups, one channel half each with its 72 KiB of weights resident,
// one DMA-staged 10 x 66 plane tile per step and one barrier per step.  The four plane steps of a tile issue 36 + 36 (f16 planes)
// and 18 + 18 (FP6 planes) MFMAs.  The epilogue of tile n runs inside tile n + 1: its residual is loaded at the top of tile n's
// last step, added (or the ReLU applied) under step 0, the conversions run under steps 1 and 2, and the eight 16-byte stores
// leave two to four per step from step 1 on.
// NOMEM (ablation builds of the micro-benchmark only, wrong results): 1 no tile DMA, 2 no MFMAs, 4 no epilogue loads / stores,
// 8 no fragment reads from LDS, 16 no epilogue arithmetic
// OUT_SPLIT (the trunk's last layer, a residual one): a.out takes split-planar records for conv2d_transpose_1.  Its epilogue is not
// pipelined -- residual, conversion and stores follow each tile's last step -- which costs one launch in twenty a few microseconds
// and keeps this variant's registers far from the limit (pipelined, with both output forms' state alive, it spilled).
template <bool RELU, bool RES, int NOMEM = 0, bool OUT_SPLIT = false>
__global__ __launch_bounds__(512, 2) void trunk6_conv_kernel(const TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wl = smem;
    unsigned char* const abuf = smem + T6::W_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int half = slot & 1, pair = slot >> 1;
    const int npairs = (int)(gridDim.x >> 4);
    const int per_img = a.tiles_x * a.tiles_y;
    const int ntiles = per_img * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int t = xcd * per_xcd + pair;
    if (t >= t_end) return;

    const unsigned hplane = (unsigned)(H * W * 32);          // one half plane (units 2 h, 2 h + 1 of a plane) of one image
    const unsigned plane_bytes = 2u * hplane;

    {
        const unsigned char* wsrc = a.wpack + (size_t)half * T6::W_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(wsrc + k * 1024), (SR_AS3 void*)(wl + k * 1024), 16, 0, 0);
        }
        if (tid < 32) reinterpret_cast<float*>(smem + T6::BIAS_OFF)[tid] = a.bias[32 * half + tid];
    }

    int kq[T6::PER_WAVE];
    int goff[T6::PER_WAVE];
    int pyx[T6::PER_WAVE];
#pragma unroll
    for (int i = 0; i < T6::PER_WAVE; ++i) {
        kq[i] = wave + 8 * i < T6::NINST ? wave + 8 * i : wave + 8 * (i - 1);
        const int u = kq[i] * 64 + lane;
        const int p = u >> 2, sl = (u & 3) ^ ((p >> 2) & 3);
        const int py = p / T6::HC, px = p - py * T6::HC;
        const bool exists = u < T6::NUNIT;
        goff[i] = (py * W + px) * 32 + (sl >> 1) * (int)hplane + (sl & 1) * 16;
        pyx[i] = exists ? (px | (py << 8)) : (1 << 28);
        asm volatile("" : "+v"(goff[i]), "+v"(pyx[i]));
    }
    const unsigned zoff = a.zero_off + (lane & 3) * 16;
    const unsigned dump = a.zero_off + 256 + lane * 32;          // lanes outside the image load / store here (64 bytes at most)
    struct Tile {
        unsigned off0;      // offset of the halo origin in unit plane 0 of plane 0
        unsigned okm;       // bit i: this lane's unit of DMA instruction i lies inside the image
        unsigned pix;       // epilogue: offset of this lane's pixel (M block h, column r) in unit plane 0 of plane 0, or ~0u outside the image
        unsigned pixs;      // OUT_SPLIT: its offset in plane 0 of the split-planar output
    };
    auto locate = [&](int tt) {
        Tile q;
        const int b = tt / per_img;
        const int rem = tt - b * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int y0 = ty * T6::TH - 1, x0 = tx * T6::TWX - 1;
        q.off0 = (unsigned)(b * 4) * plane_bytes + (unsigned)((y0 * W + x0) * 32);
        q.okm = 0;
#pragma unroll
        for (int i = 0; i < T6::PER_WAVE; ++i) {
            const unsigned gy = (unsigned)(y0 + (pyx[i] >> 8)), gx = (unsigned)(x0 + (pyx[i] & 255));
            q.okm |= ((gy < (unsigned)H) & (gx < (unsigned)W)) ? (1u << i) : 0u;
        }
        const int y = y0 + 1 + wave, x = x0 + 1 + 32 * h + r;
        q.pix = (y < H && x < W) ? (unsigned)(b * 4) * plane_bytes + (unsigned)(y * W + x) * 32u : ~0u;
        q.pixs = OUT_SPLIT ? (unsigned)(b * 4) * plane_bytes + (unsigned)(y * W + x) * 64u : 0u;
        return q;
    };
    // byte offset of unit u of plane pl of the lane's pixel, or the dump area (two 16-byte slots per lane)
    auto uoff = [&](unsigned pix, int pl, int u) { return pix != ~0u ? pix + (unsigned)(2 * pl + (u >> 1)) * hplane + 16u * (u & 1) : dump + 16u * (u & 1); };
    auto issue = [&](const Tile& q, int c, int buf) {
        if (NOMEM & 1) return;
        const unsigned offc = q.off0 + (unsigned)c * plane_bytes;
        unsigned char* dst = abuf + buf * T6::A_BYTES;
#pragma unroll
        for (int i = 0; i < T6::PER_WAVE; ++i) {
            const unsigned off = ((q.okm >> i) & 1u) ? offc + (unsigned)goff[i] : zoff;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.in + off), (SR_AS3 void*)(dst + kq[i] * 1024), 16, 0, 0);
        }
    };
    Tile cur = locate(t);
    issue(cur, 0, 0);

    auto a_addr = [&](int ky, int kx, int mb) {
        const int p = (wave + ky) * T6::HC + mb * 32 + r + kx;
        return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4);
    };
    const unsigned char* const bl = wl + lane * 16;
    const unsigned char* const bias_l = smem + T6::BIAS_OFF + 16 * h;   // channels 8 g + 4 h + (0..3) of the half

    // residual of the finished tile's pixel: its f16 record (64 B) and lo6 block (32 B).  Inline asm: see trunk_conv_kernel
    u32x4 rh[4], rl[2];
    auto res_load = [&](unsigned pix) {
        if (NOMEM & 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) rh[k] = u32x4{0u, 0u, 0u, 0u};
            rl[0] = rl[1] = u32x4{0u, 0u, 0u, 0u};
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned o = uoff(pix, half, k);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rh[k]) : "v"(o), "s"(a.res) : "memory");
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned o = uoff(pix, 3, 2 * half + k);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rl[k]) : "v"(o), "s"(a.res) : "memory");
        }
    };
    auto res_landed = [&]() { asm volatile("" : "+v"(rh[0]), "+v"(rh[1]), "+v"(rh[2]), "+v"(rh[3]), "+v"(rl[0]), "+v"(rl[1])); };
    auto store16 = [&](unsigned off, const u32x4& v) {
        if (NOMEM & 4) asm volatile("" ::"v"(v));
        else *reinterpret_cast<u32x4*>(a.out + off) = v;
    };

    f32x16 acc[2], V0, V1;      // running tile; finished tile's pixel, lane half 0 / 1 channels
    unsigned done = ~0u;        // its pixel offset
    Rec6 rec;
    f16x32 lodec;               // residual layers: the lo6 block of the finished tile's pixel, decoded (plane step 0)
    float ml = 0.f;             // running max |value - hi| of the encoder (plane step 1)
#pragma unroll
    for (int i = 0; i < 16; ++i) V0[i] = V1[i] = 0.f;
    if (RES && !OUT_SPLIT) res_load(~0u);     // the (empty) tile that drains under the first one

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // the bias
    int q = 0;
    for (;;) {
        const int tn = t + npairs;
        const bool have_next = tn < t_end;
        const Tile nxt = locate(have_next ? tn : t);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + 32 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][4 * g + j] = acc[1][4 * g + j] = b4[j];
        }

        auto plane_step = [&](auto cc) {
            constexpr int C = decltype(cc)::value;
            constexpr int NM = C < 2 ? 4 : 2;       // MFMAs per tap
            // The DMA of this step has landed; younger in the vmcnt queue: only the stores the previous step issued (x6: 2 in step 1,
            // hi: 4 in step 2, lo6: 2 in step 3).  A step's stores so have the whole
            // next step to complete, and never more than four are waited for at once: all eight in one step, waited for by the
            // next, cost 25 us of a 64 us launch (tools/probe/trunk6_ablate.py).
            constexpr int YOUNGER = (NOMEM & 4) ? 0 : OUT_SPLIT ? (C == 0 ? 8 : 0) : (C == 0 ? 2 : C == 1 ? 0 : C == 2 ? 2 : 4);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the residual of the tile being computed, a whole step before its epilogue begins (step 0 of the next tile); before the
            // DMA: older in the vmcnt queue, landed when that step has waited for its tile
            if (RES && !OUT_SPLIT && C == 3) res_load(cur.pix);
            if (RES && !OUT_SPLIT && C == 0) res_landed();
            if (C < 3) issue(cur, C + 1, (q + 1) & 1);
            else issue(nxt, 0, (q + 1) & 1);
            const unsigned char* at = abuf + (q & 1) * T6::A_BYTES;
            const unsigned char* bw = bl + C * (9 * 2 * 1024);
            ++q;
            u32x4 fa[2][2][2], fb[2][2];
            auto load_tap = [&](int tap, int s) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int ad = a_addr(ky, kx, mb);
                    fa[s][mb][0] = *reinterpret_cast<const u32x4*>(at + ad);
                    fa[s][mb][1] = *reinterpret_cast<const u32x4*>(at + (ad ^ 16));
                }
                fb[s][0] = *reinterpret_cast<const u32x4*>(bw + (tap * 2 + 0) * 1024);
                fb[s][1] = *reinterpret_cast<const u32x4*>(bw + (tap * 2 + 1) * 1024);
            };
            if (NOMEM & 8) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    fb[s2][0] = fb[s2][1] = u32x4{(unsigned)lane, 1u, 2u, 3u};
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) fa[s2][mb][0] = fa[s2][mb][1] = u32x4{(unsigned)lane, 5u, 6u, 7u};
                }
                asm volatile("" : "+v"(fb[0][0]), "+v"(fb[1][0]), "+v"(fa[0][0][0]), "+v"(fa[1][0][0]));
            } else {
                load_tap(0, 0);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = tap & 1;
                if (tap + 1 < 9 && !(NOMEM & 8)) load_tap(tap + 1, s ^ 1);
                if constexpr (NOMEM & 2) {
                    asm volatile("" ::"v"(fb[s][0]), "v"(fb[s][1]), "v"(fa[s][0][0]), "v"(fa[s][0][1]), "v"(fa[s][1][0]), "v"(fa[s][1][1]));
                } else if constexpr (C < 2) {
                    const f16x8 w0 = __builtin_bit_cast(f16x8, fb[s][0]), w1 = __builtin_bit_cast(f16x8, fb[s][1]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, __builtin_bit_cast(f16x8, fa[s][mb][0]), acc[mb], 0, 0, 0);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, __builtin_bit_cast(f16x8, fa[s][mb][1]), acc[mb], 0, 0, 0);
                    }
                } else {
                    const i32x8 wv = {(int)fb[s][0][0], (int)fb[s][0][1], (int)fb[s][0][2], (int)fb[s][0][3],
                                      (int)fb[s][1][0], (int)fb[s][1][1], (int)fb[s][1][2], (int)fb[s][1][3]};
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const i32x8 xv = {(int)fa[s][mb][0][0], (int)fa[s][mb][0][1], (int)fa[s][mb][0][2], (int)fa[s][mb][0][3],
                                          (int)fa[s][mb][1][0], (int)fa[s][mb][1][1], (int)fa[s][mb][1][2], (int)fa[s][mb][1][3]};
                        acc[mb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[mb], 2, 2, 0, (int)fb[s][1][2], 0, (int)fa[s][mb][1][2]);
                    }
                }
                // The finished tile's epilogue, spread over the steps.  Step 0: ReLU, or the residual (one decode, then a pair of
                // registers of each lane half per tap).  Step 1: the FP6 image under tap 0, two hi pairs under each other tap, the x6
                // stores.  Step 2: the FP6 residue under tap 0, the hi stores.  Step 3: the lo6 stores.
                if (RELU && !OUT_SPLIT && C == 0 && tap >= 1 && !(NOMEM & 16)) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int i = 2 * (tap - 1) + k;
                        V0[i] = __builtin_amdgcn_fmed3f(V0[i], 0.f, __builtin_inff());
                        V1[i] = __builtin_amdgcn_fmed3f(V1[i], 0.f, __builtin_inff());
                    }
                }
                if constexpr (RES && !OUT_SPLIT && C == 0 && !(NOMEM & 16)) {
                    if (tap == 0) lodec = dec_lo(rl);
                    else dec_add_pair(V0, V1, tap - 1, rh, lodec);
                }
                if constexpr (OUT_SPLIT) {
                } else if constexpr (NOMEM & 16) {
                    const u32x4 junk = u32x4{(unsigned)lane, 0u, 0u, 0u};
                    if ((C == 1 && tap == 3) || (C == 2 && (tap == 2 || tap == 3)) || (C == 3 && tap == 2)) {
                        store16(uoff(done, half, 0), junk);
                        store16(uoff(done, half, 1), junk);
                    }
                } else {
                    if constexpr (C == 1) {
                        if (tap == 0) {
                            enc_x6(V0, V1, rec);
                            ml = 0.f;
                        }
                        if (tap == 1) { enc_hi_pairs<0>(V0, V1, rec, ml); enc_hi_pairs<1>(V0, V1, rec, ml); }
                        if (tap == 2) { enc_hi_pairs<2>(V0, V1, rec, ml); enc_hi_pairs<3>(V0, V1, rec, ml); }
                        if (tap == 3) { enc_hi_pairs<4>(V0, V1, rec, ml); enc_hi_pairs<5>(V0, V1, rec, ml); }
                        if (tap == 4) { enc_hi_pairs<6>(V0, V1, rec, ml); enc_hi_pairs<7>(V0, V1, rec, ml); }
                        if (tap == 5) { enc_hi_pairs<8>(V0, V1, rec, ml); enc_hi_pairs<9>(V0, V1, rec, ml); }
                        if (tap == 6) { enc_hi_pairs<10>(V0, V1, rec, ml); enc_hi_pairs<11>(V0, V1, rec, ml); }
                        if (tap == 7) { enc_hi_pairs<12>(V0, V1, rec, ml); enc_hi_pairs<13>(V0, V1, rec, ml); }
                        if (tap == 8) { enc_hi_pairs<14>(V0, V1, rec, ml); enc_hi_pairs<15>(V0, V1, rec, ml); }
                        if (tap == 3) {
                            store16(uoff(done, 2, 2 * half), rec.x6[0]);
                            store16(uoff(done, 2, 2 * half + 1), rec.x6[1]);
                        }
                    }
                    if constexpr (C == 2) {
                        if (tap == 0) enc_lo6(V0, V1, ml, rec);
                        if (tap == 2 || tap == 3) {
                            store16(uoff(done, half, 2 * (tap - 2)), rec.hi[2 * (tap - 2)]);
                            store16(uoff(done, half, 2 * (tap - 2) + 1), rec.hi[2 * (tap - 2) + 1]);
                        }
                    }
                    if constexpr (C == 3) {
                        if (tap == 2) {
                            store16(uoff(done, 3, 2 * half), rec.lo6[0]);
                            store16(uoff(done, 3, 2 * half + 1), rec.lo6[1]);
                        }
                    }
                }
                // behind each MFMA: the next tap's fragment reads and a share of the VALU / DMA instructions
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (NM == 4) {
                        if (tap + 1 < 9) {
                            if (k < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        if (tap == 0) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
                    } else {
                        if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        if (tap == 0) __builtin_amdgcn_sched_group_barrier(0x010, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 16, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        plane_step(std::integral_constant<int, 0>{});
        plane_step(std::integral_constant<int, 1>{});
        plane_step(std::integral_constant<int, 2>{});
        plane_step(std::integral_constant<int, 3>{});
        // lane (r, h) takes all 32 channels of pixel (M block h, column r): V0 = lane half 0's registers, V1 = lane half 1's
        swap_halves(acc[0], acc[1], V0, V1);
        done = cur.pix;
        if constexpr (OUT_SPLIT) {      // this tile's epilogue, here and now
            res_load(done);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the next tile's first DMA with it)
            res_landed();
            decode_add6(V0, V1, rh, rl);
            const unsigned d0 = done != ~0u ? cur.pixs + (unsigned)(2 * half) * plane_bytes : dump;
            const unsigned d1 = done != ~0u ? d0 + plane_bytes : dump;
            u32x4 o[4];
            encode_split_plane<0>(V0, V1, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) store16(d0 + (done != ~0u ? 16u * k : 16u * (k & 1)), o[k]);
            encode_split_plane<1>(V0, V1, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) store16(d1 + (done != ~0u ? 16u * k : 16u * (k & 1)), o[k]);
        }
        if (!have_next) break;
        t = tn;
        cur = nxt;
    }
    if constexpr (OUT_SPLIT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // the last tile's epilogue has no MFMAs to hide under
    if (RES) {      // its residual came with the last plane step
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        res_landed();
        decode_add6(V0, V1, rh, rl);
    }
    if (RELU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            V0[i] = __builtin_amdgcn_fmed3f(V0[i], 0.f, __builtin_inff());
            V1[i] = __builtin_amdgcn_fmed3f(V1[i], 0.f, __builtin_inff());
        }
    }
    {
        encode6(V0, V1, rec);
#pragma unroll
        for (int k = 0; k < 4; ++k) store16(uoff(done, half, k), rec.hi[k]);
        store16(uoff(done, 2, 2 * half), rec.x6[0]);
        store16(uoff(done, 2, 2 * half + 1), rec.x6[1]);
        store16(uoff(done, 3, 2 * half), rec.lo6[0]);
        store16(uoff(done, 3, 2 * half + 1), rec.lo6[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// byte offset of unit u of plane `plane` of pixel p of image b in an mx6 tensor
__device__ __forceinline__ size_t mx6_unit(long b, int HW, int plane, int u, long p) {
    return (((size_t)b * 8 + 2 * plane + (u >> 1)) * HW + p) * 32 + 16 * (u & 1);
}

// the records of one pixel's 32-channel block blk to their unit planes
__device__ __forceinline__ void mx6_store(unsigned char* out, long b, int HW, long p, int blk, const Rec6& o) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(out + mx6_unit(b, HW, blk, k, p)) = o.hi[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        *reinterpret_cast<u32x4*>(out + mx6_unit(b, HW, 2, 2 * blk + k, p)) = o.x6[k];
        *reinterpret_cast<u32x4*>(out + mx6_unit(b, HW, 3, 2 * blk + k, p)) = o.lo6[k];
    }
}

// ---- fp32 NHWC (64 channels) <-> mx6 / split-planar, for the per-op entry point sr_op_conv3x3_mx6 and its tests ----------------------

__global__ __launch_bounds__(256) void nhwc_to_mx6_kernel(const float* x, unsigned char* out, int B, int HW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * HW * 2) return;
    const int blk = (int)(idx & 1);
    const long pb = idx >> 1;
    const long b = pb / HW, p = pb - b * HW;
    const float* src = x + (size_t)pb * 64 + 32 * blk;
    f32x16 V0, V1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        V0[i] = src[(i & 3) + 8 * (i >> 2)];
        V1[i] = src[(i & 3) + 8 * (i >> 2) + 4];
    }
    Rec6 o;
    encode6(V0, V1, o);
    mx6_store(out, b, HW, p, blk, o);
}

// which = 0: the stored value hi + lo6; 1: the x6 plane alone (what the third MFMA term reads)
__global__ __launch_bounds__(256) void mx6_to_nhwc_kernel(const unsigned char* in, float* y, int B, int HW, int which) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * HW * 2) return;
    const int blk = (int)(idx & 1);
    const long pb = idx >> 1;
    const long b = pb / HW, p = pb - b * HW;
    u32x4 rh[4], rl[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) rh[k] = which ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4*>(in + mx6_unit(b, HW, blk, k, p));
#pragma unroll
    for (int k = 0; k < 2; ++k) rl[k] = *reinterpret_cast<const u32x4*>(in + mx6_unit(b, HW, which ? 2 : 3, 2 * blk + k, p));
    f32x16 V0, V1;
#pragma unroll
    for (int i = 0; i < 16; ++i) V0[i] = V1[i] = 0.f;
    decode_add6(V0, V1, rh, rl);
    float* dst = y + (size_t)pb * 64 + 32 * blk;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        dst[(i & 3) + 8 * (i >> 2)] = V0[i];
        dst[(i & 3) + 8 * (i >> 2) + 4] = V1[i];
    }
}

__global__ __launch_bounds__(256) void split_to_nhwc_kernel(const unsigned char* in, float* y, int B, int HW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * HW * 4) return;
    const int pl = (int)(idx & 3);
    const long pb = idx >> 2;
    const long b = pb / HW, p = pb - b * HW;
    const size_t plane = (size_t)HW * 64;
    const u32x4* rec = reinterpret_cast<const u32x4*>(in + ((size_t)b * 4 + pl) * plane + (size_t)p * 64);
    float* dst = y + (size_t)pb * 64 + 16 * pl;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const u32x4 hv = rec[2 * hh], lv = rec[2 * hh + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[split_perm(hh, 2 * k)] = __builtin_bit_cast(float, hv[k] << 16) + __builtin_bit_cast(float, lv[k] << 16);
            dst[split_perm(hh, 2 * k + 1)] = __builtin_bit_cast(float, hv[k] & 0xFFFF0000u) + __builtin_bit_cast(float, lv[k] & 0xFFFF0000u);
        }
    }
}

hipError_t launch_nhwc_to_mx6(const float* x, void* out, int B, int H, int W, hipStream_t s) {
    const long n = (long)B * H * W * 2;
    hipLaunchKernelGGL(nhwc_to_mx6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (unsigned char*)out, B, H * W);
    return hipGetLastError();
}
hipError_t launch_mx6_to_nhwc(const void* in, float* y, int B, int H, int W, int which, hipStream_t s) {
    const long n = (long)B * H * W * 2;
    hipLaunchKernelGGL(mx6_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned char*)in, y, B, H * W, which);
    return hipGetLastError();
}
hipError_t launch_split_to_nhwc(const void* in, float* y, int B, int H, int W, hipStream_t s) {
    const long n = (long)B * H * W * 4;
    hipLaunchKernelGGL(split_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned char*)in, y, B, H * W);
    return hipGetLastError();
}

static int trunk6_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <bool RELU, bool RES, int NOMEM = 0, bool OUT_SPLIT = false>
static hipError_t launch_trunk6_t(const TrunkArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trunk6_conv_kernel<RELU, RES, NOMEM, OUT_SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T6::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((trunk6_conv_kernel<RELU, RES, NOMEM, OUT_SPLIT>), dim3(grid), dim3(512), T6::LDS, s, a);
    return hipGetLastError();
}

// a.in / a.res mx6, a.out mx6 or (out_split, residual layer, NOT in place) split-planar; ReLU layers and residual layers like
// launch_trunk_conv
hipError_t launch_trunk6_conv(TrunkArgs a, int out_split, hipStream_t s) {
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.out || !a.wpack || !a.bias) return hipErrorInvalidValue;
    if ((double)a.B * a.H * a.W * 256.0 + 4096.0 > 4294967295.0 || (double)a.zero_off + 4096.0 > 4294967295.0) return hipErrorInvalidValue;
    if ((a.relu != 0) == (a.res != nullptr)) return hipErrorInvalidValue;
    a.tiles_x = (a.W + T6::TWX - 1) / T6::TWX;
    a.tiles_y = (a.H + T6::TH - 1) / T6::TH;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    int npairs = std::max(1, trunk6_num_cus() / 16);
    npairs = std::min(npairs, per_xcd);
    const int rounds = (per_xcd + npairs - 1) / npairs;
    npairs = (per_xcd + rounds - 1) / rounds;
    const int grid = 16 * npairs;
    if (out_split) return a.res ? launch_trunk6_t<false, true, 0, true>(a, grid, s) : hipErrorInvalidValue;   // the trunk's last layer only
    if (a.dbg && a.res) {      // micro-benchmark ablations (tools/probe/trunk_ablate_b.py), residual layer
        switch (a.dbg) {
            case 1: return launch_trunk6_t<false, true, 1>(a, grid, s);
            case 4: return launch_trunk6_t<false, true, 4>(a, grid, s);
            case 5: return launch_trunk6_t<false, true, 5>(a, grid, s);
            case 16: return launch_trunk6_t<false, true, 16>(a, grid, s);
            case 21: return launch_trunk6_t<false, true, 21>(a, grid, s);      // MFMAs + fragment reads only
            case 29: return launch_trunk6_t<false, true, 29>(a, grid, s);      // MFMAs only
            case 23: return launch_trunk6_t<false, true, 23>(a, grid, s);      // fragment reads only
            case 30: return launch_trunk6_t<false, true, 30>(a, grid, s);      // tile DMA only
            case 27: return launch_trunk6_t<false, true, 27>(a, grid, s);      // epilogue memory only
            default: return launch_trunk6_t<false, true, 31>(a, grid, s);      // the empty pipeline
        }
    }
    return a.res ? launch_trunk6_t<false, true>(a, grid, s) : launch_trunk6_t<true, false>(a, grid, s);
}

// ---- host side: the packed weights of one 64 -> 64 layer, w = Keras kernel (3, 3, 64, 64) --------------------------------------------

static uint16_t f32_to_f16_bits(float f) {
    const _Float16 hv = (_Float16)f;      // round to nearest even
    uint16_t u;
    memcpy(&u, &hv, 2);
    return u;
}
static float f16_bits_to_f32(uint16_t u) {
    _Float16 hv;
    memcpy(&hv, &u, 2);
    return (float)hv;
}
static float mx_scale_host(float m) {
    float t = m * 0.13333334f;
    uint32_t b;
    memcpy(&b, &t, 4);
    b = (b + 0x7FFFFFu) & 0x7F800000u;
    if (b < 0x0C800000u) b = 0x0C800000u;
    memcpy(&t, &b, 4);
    return t;
}
// E2M3 code of q (|q| <= 7.5 after the scale), round to nearest even
static unsigned fp6_code(float q) {
    const float a = std::fabs(q);
    int code;
    if (a < 2.f) code = (int)std::nearbyint(a * 8.f);
    else if (a < 4.f) code = 8 + (int)std::nearbyint(a * 4.f);
    else code = 16 + (int)std::nearbyint(a * 2.f);
    if (code > 31) code = 31;
    return (unsigned)code | (q < 0.f ? 32u : 0u);
}
static int c6(int e) { return ((e >> 1) & 3) + 8 * (e >> 3) + 4 * (e & 1); }

size_t trunk6_weight_bytes() { return (size_t)2 * T6::W_BYTES; }

// out[trunk6_weight_bytes()]: [cout half][step 0, 1: f16 hi of input plane P; step 2: fp6(w - f16 w); step 3: fp6(w)][tap][2 units][lane] x 16 B
void trunk6_pack_weights(const float* w, unsigned char* out) {
    memset(out, 0, trunk6_weight_bytes());
    auto W = [&](int tap, int ci, int co) { return w[((size_t)tap * 64 + ci) * 64 + co]; };
    for (int hf = 0; hf < 2; ++hf) {
        unsigned char* base = out + (size_t)hf * T6::W_BYTES;
        for (int tap = 0; tap < 9; ++tap)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 31, hh = lane >> 5, co = 32 * hf + j;
                for (int P = 0; P < 2; ++P)
                    for (int ks = 0; ks < 2; ++ks) {
                        unsigned char* u = base + ((size_t)(P * 9 + tap) * 2 + ks) * 1024 + lane * 16;
                        for (int e = 0; e < 8; ++e) {
                            const uint16_t v = f32_to_f16_bits(W(tap, 32 * P + 16 * ks + split_perm(hh, e), co));
                            memcpy(u + 2 * e, &v, 2);
                        }
                    }
                for (int term = 0; term < 2; ++term) {      // 0: the low part (step 2), 1: the value (step 3)
                    float v[32], m = 0.f;
                    for (int e = 0; e < 32; ++e) {
                        const float x = W(tap, 32 * hh + c6(e), co);
                        v[e] = term ? x : x - f16_bits_to_f32(f32_to_f16_bits(x));
                        m = std::max(m, std::fabs(v[e]));
                    }
                    const float sc = mx_scale_host(m);
                    uint32_t sb;
                    memcpy(&sb, &sc, 4);
                    unsigned char bytes[32] = {};
                    for (int e = 0; e < 32; ++e) {
                        const unsigned code = fp6_code(v[e] / sc);
                        const int bit = 6 * e;
                        bytes[bit >> 3] |= (unsigned char)(code << (bit & 7));
                        if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
                    }
                    bytes[24] = (unsigned char)(sb >> 23);
                    unsigned char* u = base + ((size_t)((2 + term) * 9 + tap) * 2) * 1024 + lane * 16;
                    memcpy(u, bytes, 16);
                    memcpy(u + 1024, bytes + 16, 16);
                }
            }
    }
}

}  // namespace sr
