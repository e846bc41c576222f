// SRNet trunk, second arithmetic: the 20 residual-block convs (reference layers.py:58-62) with the fp32 product split as
//     w * x  ~=  f16(w) * f16(x)  +  fp6(w) * fp6(x - f16(x))  +  fp6(w - f16(w)) * fp6(x)
// one v_mfma_f32_32x32x16_f16 per 16 channels for the first term and one MX-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (FP6 E2M3,
// one E8M0 scale per 32 channels) per 64 channels for each cross term: 6 MFMA issues per tap and 64 channels where the
// bf16x3 kernel of conv_trunk.hip needs 12, at the same 32 cycles each (tools/probe/mx_rate_probe.hip).  The cross terms are
// 2^-11 of the product, so 4 significant bits on them leave the sum at ~2^-15 relative: measured against the fp64 oracle in
// tests/test_gpu_mx6.py (bar 1e-3 on [0, 1] pixels).
//
// Data layout ("mx6"): the same geometry as split-planar, [B][4 planes][H][W][64 B], so the tile DMA, the swizzle and every LDS
// read address of conv_trunk.hip carry over unchanged.  Per pixel:
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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define SR_AS1 __attribute__((address_space(1)))
#define SR_AS3 __attribute__((address_space(3)))

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

// power of two s with m / s <= 7.5 (the largest E2M3 value); its exponent field is the E8M0 scale byte
__device__ __forceinline__ unsigned mx_scale_bits(float m) {
    unsigned b = (__builtin_bit_cast(unsigned, m * 0.13333334f) + 0x7FFFFFu) & 0x7F800000u;
    return b < 0x0C800000u ? 0x0C800000u : b;      // >= 2^-102: m = 0 (or denormal) still divides
}

__device__ __forceinline__ i32x6 cvt_fp6_2x16(const f32x16& a, const f32x16& b, float scale) {
    return __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);      // element 2 i = a[i] / scale, 2 i + 1 = b[i] / scale
}
__device__ __forceinline__ f16x32 cvt_f16_fp6_32(const i32x6& d, float scale) {
    return __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(d, scale);           // element e * scale
}

// the records of one pixel's 32 channels: V0 / V1 = the accumulator registers of lane half 0 / 1
struct Rec6 {
    u32x4 hi[4];    // f16 plane record: units 0, 1 from V0, units 2, 3 from V1
    u32x4 x6[2];    // 24 bytes of FP6 + scale byte at byte 24
    u32x4 lo6[2];
};

// encode6 in the three stages the kernel spreads over a tile's plane steps: the FP6 image of the values, then pair by pair the f16
// hi parts (V becomes value - hi in place, ml its running maximum), then the FP6 image of what is left
__device__ __forceinline__ void enc_x6(const f32x16& V0, const f32x16& V1, Rec6& o) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fmaxf(fabsf(V0[i]), fabsf(V1[i])));
    const unsigned sx = mx_scale_bits(mx);
    const i32x6 cx = cvt_fp6_2x16(V0, V1, __builtin_bit_cast(float, sx));
    o.x6[0] = u32x4{(unsigned)cx[0], (unsigned)cx[1], (unsigned)cx[2], (unsigned)cx[3]};
    o.x6[1] = u32x4{(unsigned)cx[4], (unsigned)cx[5], sx >> 23, 0u};
}
template <int HH, int I>
__device__ __forceinline__ void enc_hi_pair(f32x16& V, Rec6& o, float& ml) {
    // f16 keeps 65504: what lies above goes to the FP6 parts instead of becoming inf
    const float a = __builtin_amdgcn_fmed3f(V[2 * I], -65504.f, 65504.f), b = __builtin_amdgcn_fmed3f(V[2 * I + 1], -65504.f, 65504.f);
    const f16x2 p = {(_Float16)a, (_Float16)b};
    o.hi[2 * HH + (I >> 2)][I & 3] = __builtin_bit_cast(unsigned, p);
    V[2 * I] -= (float)p[0];
    V[2 * I + 1] -= (float)p[1];
    ml = fmaxf(ml, fmaxf(fabsf(V[2 * I]), fabsf(V[2 * I + 1])));
}
__device__ __forceinline__ void enc_lo6(const f32x16& V0, const f32x16& V1, float ml, Rec6& o) {
    const unsigned sl = mx_scale_bits(ml);
    const i32x6 cl = cvt_fp6_2x16(V0, V1, __builtin_bit_cast(float, sl));
    o.lo6[0] = u32x4{(unsigned)cl[0], (unsigned)cl[1], (unsigned)cl[2], (unsigned)cl[3]};
    o.lo6[1] = u32x4{(unsigned)cl[4], (unsigned)cl[5], sl >> 23, 0u};
}
template <int N>
__device__ __forceinline__ void enc_hi_pairs(f32x16& V0, f32x16& V1, Rec6& o, float& ml) {      // pair N of the 16: half N >> 3, pair N & 7
    if constexpr (N < 8) enc_hi_pair<0, N>(V0, o, ml);
    else enc_hi_pair<1, N - 8>(V1, o, ml);
}
template <int N>
struct EncLoop {
    static __device__ __forceinline__ void run(f32x16& V0, f32x16& V1, Rec6& o, float& ml) {
        EncLoop<N - 1>::run(V0, V1, o, ml);
        enc_hi_pairs<N - 1>(V0, V1, o, ml);
    }
};
template <>
struct EncLoop<0> {
    static __device__ __forceinline__ void run(f32x16&, f32x16&, Rec6&, float&) {}
};
__device__ __forceinline__ void encode6(f32x16& V0, f32x16& V1, Rec6& o) {
    float ml = 0.f;
    enc_x6(V0, V1, o);
    EncLoop<16>::run(V0, V1, o, ml);
    enc_lo6(V0, V1, ml, o);
}

// the two f16 of a dword as floats, from a scalar: __builtin_bit_cast applied to a vector ELEMENT (bit_cast<f16x2>(v[k])) reads
// element 0 whatever k is (hipcc, ROCm 7.2; tools/probe/mx6_dbg.py)
__device__ __forceinline__ float h_even(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)); }
__device__ __forceinline__ float h_odd(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }

// V += hi + lo6 of a stored pixel (rh = its f16 record, rl = its lo6 block), in two stages: the FP6 block as 32 f16, then pair by pair
__device__ __forceinline__ f16x32 dec_lo(const u32x4 (&rl)[2]) {
    const i32x6 d = {(int)rl[0][0], (int)rl[0][1], (int)rl[0][2], (int)rl[0][3], (int)rl[1][0], (int)rl[1][1]};
    const float s = __builtin_bit_cast(float, (rl[1][2] & 0xFFu) << 23);
    return cvt_f16_fp6_32(d, s);
}
__device__ __forceinline__ void dec_add_pair(f32x16& V0, f32x16& V1, int i, const u32x4 (&rh)[4], const f16x32& lo) {
    const unsigned p0 = rh[i >> 2][i & 3], p1 = rh[2 + (i >> 2)][i & 3];
    V0[2 * i] += h_even(p0) + (float)lo[4 * i];
    V1[2 * i] += h_even(p1) + (float)lo[4 * i + 1];
    V0[2 * i + 1] += h_odd(p0) + (float)lo[4 * i + 2];
    V1[2 * i + 1] += h_odd(p1) + (float)lo[4 * i + 3];
}
__device__ __forceinline__ void decode_add6(f32x16& V0, f32x16& V1, const u32x4 (&rh)[4], const u32x4 (&rl)[2]) {
    const f16x32 lo = dec_lo(rl);
#pragma unroll
    for (int i = 0; i < 8; ++i) dec_add_pair(V0, V1, i, rh, lo);
}

// the same 32 channels as two split-planar (conv_trunk.hip) plane records: what conv2d_transpose_1 reads
__device__ __forceinline__ void encode_split(const f32x16& V0, const f32x16& V1, u32x4 (&o)[8]) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x16& V = hh ? V1 : V0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = V[8 * pl + 2 * k], b = V[8 * pl + 2 * k + 1];
                const bf16x2 hv = {(__bf16)a, (__bf16)b};
                const unsigned hu = __builtin_bit_cast(unsigned, hv);
                const bf16x2 lv = {(__bf16)(a - __builtin_bit_cast(float, hu << 16)), (__bf16)(b - __builtin_bit_cast(float, hu & 0xFFFF0000u))};
                o[4 * pl + 2 * hh][k] = hu;
                o[4 * pl + 2 * hh + 1][k] = __builtin_bit_cast(unsigned, lv);
            }
        }
}

}  // namespace

// RELU: ReLU after the bias; RES: add a.res (mx6) after it.  OUT_SPLIT: write split-planar records (the last trunk layer).
//
// Same skeleton as trunk_conv_kernel: persistent 512-thread workgroups, one channel half each with its 72 KiB of weights resident,
// one DMA-staged 10 x 66 plane tile per step and one barrier per step.  The four plane steps of a tile issue 36 + 36 (f16 planes)
// and 18 + 18 (FP6 planes) MFMAs.  The epilogue of tile n runs inside tile n + 1: its residual is loaded at the top of tile n's
// last step, added (or the ReLU applied) under step 0, the conversions run under steps 1 and 2, and the eight 16-byte stores
// leave two to four per step from step 1 on.
// NOMEM (ablation builds of the micro-benchmark only, wrong results): 1 no tile DMA, 2 no MFMAs, 4 no epilogue loads / stores,
// 8 no fragment reads from LDS, 16 no epilogue arithmetic
template <bool RELU, bool RES, bool OUT_SPLIT, int NOMEM = 0>
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

    const unsigned plane_bytes = (unsigned)(H * W * 64);

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
        goff[i] = (py * W + px) * 64 + sl * 16;
        pyx[i] = exists ? (px | (py << 8)) : (1 << 28);
        asm volatile("" : "+v"(goff[i]), "+v"(pyx[i]));
    }
    const unsigned zoff = a.zero_off + (lane & 3) * 16;
    const unsigned dump = a.zero_off + 256 + lane * 32;          // lanes outside the image load / store here (64 bytes at most)
    struct Tile {
        unsigned off0;      // offset of the halo origin of plane 0
        unsigned okm;       // bit i: this lane's unit of DMA instruction i lies inside the image
        unsigned pix;       // epilogue: offset of this lane's pixel (M block h, column r) in plane 0, or ~0u outside the image
    };
    auto locate = [&](int tt) {
        Tile q;
        const int b = tt / per_img;
        const int rem = tt - b * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int y0 = ty * T6::TH - 1, x0 = tx * T6::TWX - 1;
        q.off0 = (unsigned)(b * 4) * plane_bytes + (unsigned)((y0 * W + x0) * 64);
        q.okm = 0;
#pragma unroll
        for (int i = 0; i < T6::PER_WAVE; ++i) {
            const unsigned gy = (unsigned)(y0 + (pyx[i] >> 8)), gx = (unsigned)(x0 + (pyx[i] & 255));
            q.okm |= ((gy < (unsigned)H) & (gx < (unsigned)W)) ? (1u << i) : 0u;
        }
        const int y = y0 + 1 + wave, x = x0 + 1 + 32 * h + r;
        q.pix = (y < H && x < W) ? (unsigned)(b * 4) * plane_bytes + (unsigned)(y * W + x) * 64u : ~0u;
        return q;
    };
    // byte offset of (plane pl, byte `add`) of the lane's pixel, or the dump area
    auto eoff = [&](unsigned pix, int pl, unsigned add) { return pix != ~0u ? pix + (unsigned)pl * plane_bytes + add : dump; };
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
        const unsigned oh = eoff(pix, half, 0u), ol = eoff(pix, 3, 32u * half);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rh[0]) : "v"(oh), "s"(a.res) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(rh[1]) : "v"(oh), "s"(a.res) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(rh[2]) : "v"(oh), "s"(a.res) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(rh[3]) : "v"(oh), "s"(a.res) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rl[0]) : "v"(ol), "s"(a.res) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(rl[1]) : "v"(ol), "s"(a.res) : "memory");
    };
    auto res_landed = [&]() { asm volatile("" : "+v"(rh[0]), "+v"(rh[1]), "+v"(rh[2]), "+v"(rh[3]), "+v"(rl[0]), "+v"(rl[1])); };
    auto store16 = [&](unsigned off, const u32x4& v) {
        if (NOMEM & 4) asm volatile("" ::"v"(v));
        else *reinterpret_cast<u32x4*>(a.out + off) = v;
    };

    f32x16 acc[2], V0, V1;      // running tile; finished tile's pixel, lane half 0 / 1 channels
    unsigned done = ~0u;        // its pixel offset
    Rec6 rec;
    u32x4 osp[8];
    f16x32 lodec;               // residual layers: the lo6 block of the finished tile's pixel, decoded (plane step 0)
    float ml = 0.f;             // running max |value - hi| of the encoder (plane step 1)
#pragma unroll
    for (int i = 0; i < 16; ++i) V0[i] = V1[i] = 0.f;
    if (RES) res_load(~0u);     // the (empty) tile that drains under the first one

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
            // hi: 4 in step 2, lo6: 2 in step 3; split-planar output: 4 in step 2, 4 in step 3).  A step's stores so have the whole
            // next step to complete, and never more than four are waited for at once: all eight in one step, waited for by the
            // next, cost 25 us of a 64 us launch (tools/probe/trunk6_ablate.py).
            constexpr int YOUNGER = (NOMEM & 4) ? 0 : OUT_SPLIT ? (C == 0 || C == 3 ? 4 : 0) : (C == 0 ? 2 : C == 1 ? 0 : C == 2 ? 2 : 4);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the residual of the tile being computed, a whole step before its epilogue begins (step 0 of the next tile); before the
            // DMA: older in the vmcnt queue, landed when that step has waited for its tile
            if (RES && C == 3) res_load(cur.pix);
            if (RES && C == 0) res_landed();
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
                if (RELU && C == 0 && tap >= 1 && !(NOMEM & 16)) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int i = 2 * (tap - 1) + k;
                        V0[i] = __builtin_amdgcn_fmed3f(V0[i], 0.f, __builtin_inff());
                        V1[i] = __builtin_amdgcn_fmed3f(V1[i], 0.f, __builtin_inff());
                    }
                }
                if constexpr (RES && C == 0 && !(NOMEM & 16)) {
                    if (tap == 0) lodec = dec_lo(rl);
                    else dec_add_pair(V0, V1, tap - 1, rh, lodec);
                }
                if constexpr (OUT_SPLIT) {
                    if (C == 1 && tap == 1) encode_split(V0, V1, osp);
                    if ((C == 2 || C == 3) && (tap == 2 || tap == 3)) {
                        const unsigned o = eoff(done, 2 * half + (C - 2), 0u) + 32u * (tap - 2);
                        store16(o, osp[4 * (C - 2) + 2 * (tap - 2)]);
                        store16(o + 16, osp[4 * (C - 2) + 2 * (tap - 2) + 1]);
                    }
                } else if constexpr (NOMEM & 16) {
                    const unsigned o = eoff(done, half, 0u);
                    const u32x4 junk = u32x4{(unsigned)lane, 0u, 0u, 0u};
                    if ((C == 1 && tap == 3) || (C == 2 && (tap == 2 || tap == 3)) || (C == 3 && tap == 2)) {
                        store16(o, junk);
                        store16(o + 16, junk);
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
                            const unsigned ox = eoff(done, 2, 32u * half);
                            store16(ox, rec.x6[0]);
                            store16(ox + 16, rec.x6[1]);
                        }
                    }
                    if constexpr (C == 2) {
                        if (tap == 0) enc_lo6(V0, V1, ml, rec);
                        if (tap == 2 || tap == 3) {
                            const unsigned oh = eoff(done, half, 0u) + 32u * (tap - 2);
                            store16(oh, rec.hi[2 * (tap - 2)]);
                            store16(oh + 16, rec.hi[2 * (tap - 2) + 1]);
                        }
                    }
                    if constexpr (C == 3) {
                        if (tap == 2) {
                            const unsigned ol = eoff(done, 3, 32u * half);
                            store16(ol, rec.lo6[0]);
                            store16(ol + 16, rec.lo6[1]);
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
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index (hipcc, ROCm 7.2)
            const float a0 = acc[0][i], a1 = acc[1][i];
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1), false, false);
            const unsigned s0 = sw[0], s1 = sw[1];
            V0[i] = __builtin_bit_cast(float, s0);
            V1[i] = __builtin_bit_cast(float, s1);
        }
        done = cur.pix;
        if (!have_next) break;
        t = tn;
        cur = nxt;
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
    if constexpr (OUT_SPLIT) {
        encode_split(V0, V1, osp);
        const unsigned o0 = eoff(done, 2 * half, 0u), o1 = eoff(done, 2 * half + 1, 0u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            store16(o0 + 16 * k, osp[k]);
            store16(o1 + 16 * k, osp[4 + k]);
        }
    } else {
        encode6(V0, V1, rec);
        const unsigned oh = eoff(done, half, 0u), ox = eoff(done, 2, 32u * half), ol = eoff(done, 3, 32u * half);
#pragma unroll
        for (int k = 0; k < 4; ++k) store16(oh + 16 * k, rec.hi[k]);
        store16(ox, rec.x6[0]);
        store16(ox + 16, rec.x6[1]);
        store16(ol, rec.lo6[0]);
        store16(ol + 16, rec.lo6[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// split-planar (bf16 hi / lo, conv_trunk.hip) -> mx6, one thread per (pixel, 32-channel block): the head conv's output
__global__ __launch_bounds__(256) void split_to_mx6_kernel(const unsigned char* in, unsigned char* out, int B, int HW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)B * HW * 2;
    if (idx >= n) return;
    const int blk = (int)(idx & 1);
    const long pb = idx >> 1;
    const long b = pb / HW, p = pb - b * HW;
    const size_t plane = (size_t)HW * 64;
    const unsigned char* src = in + (size_t)b * 4 * plane + (size_t)p * 64;
    f32x16 V0, V1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const u32x4* rec = reinterpret_cast<const u32x4*>(src + (size_t)(2 * blk + ks) * plane);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const u32x4 hv = rec[2 * hh], lv = rec[2 * hh + 1];
            f32x16& V = hh ? V1 : V0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                V[8 * ks + 2 * k] = __builtin_bit_cast(float, hv[k] << 16) + __builtin_bit_cast(float, lv[k] << 16);
                V[8 * ks + 2 * k + 1] = __builtin_bit_cast(float, hv[k] & 0xFFFF0000u) + __builtin_bit_cast(float, lv[k] & 0xFFFF0000u);
            }
        }
    }
    Rec6 o;
    encode6(V0, V1, o);
    unsigned char* dst = out + (size_t)b * 4 * plane + (size_t)p * 64;
    u32x4* dh = reinterpret_cast<u32x4*>(dst + (size_t)blk * plane);
#pragma unroll
    for (int k = 0; k < 4; ++k) dh[k] = o.hi[k];
    u32x4* dx = reinterpret_cast<u32x4*>(dst + 2 * plane + 32 * blk);
    dx[0] = o.x6[0];
    dx[1] = o.x6[1];
    u32x4* dl = reinterpret_cast<u32x4*>(dst + 3 * plane + 32 * blk);
    dl[0] = o.lo6[0];
    dl[1] = o.lo6[1];
}

hipError_t launch_split_to_mx6(const void* in, void* out, int B, int H, int W, hipStream_t s) {
    if (!in || !out || B < 1 || H < 1 || W < 1) return hipErrorInvalidValue;
    const long n = (long)B * H * W * 2;
    hipLaunchKernelGGL(split_to_mx6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const unsigned char*)in, (unsigned char*)out, B, H * W);
    return hipGetLastError();
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
    const size_t plane = (size_t)HW * 64;
    unsigned char* dst = out + (size_t)b * 4 * plane + (size_t)p * 64;
    u32x4* dh = reinterpret_cast<u32x4*>(dst + (size_t)blk * plane);
#pragma unroll
    for (int k = 0; k < 4; ++k) dh[k] = o.hi[k];
    u32x4* dx = reinterpret_cast<u32x4*>(dst + 2 * plane + 32 * blk);
    dx[0] = o.x6[0];
    dx[1] = o.x6[1];
    u32x4* dl = reinterpret_cast<u32x4*>(dst + 3 * plane + 32 * blk);
    dl[0] = o.lo6[0];
    dl[1] = o.lo6[1];
}

// which = 0: the stored value hi + lo6; 1: the x6 plane alone (what the third MFMA term reads)
__global__ __launch_bounds__(256) void mx6_to_nhwc_kernel(const unsigned char* in, float* y, int B, int HW, int which) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * HW * 2) return;
    const int blk = (int)(idx & 1);
    const long pb = idx >> 1;
    const long b = pb / HW, p = pb - b * HW;
    const size_t plane = (size_t)HW * 64;
    const unsigned char* src = in + (size_t)b * 4 * plane + (size_t)p * 64;
    u32x4 rh[4], rl[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) rh[k] = which ? u32x4{0u, 0u, 0u, 0u} : reinterpret_cast<const u32x4*>(src + (size_t)blk * plane)[k];
    rl[0] = reinterpret_cast<const u32x4*>(src + (which ? 2 : 3) * plane + 32 * blk)[0];
    rl[1] = reinterpret_cast<const u32x4*>(src + (which ? 2 : 3) * plane + 32 * blk)[1];
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

template <bool RELU, bool RES, bool OUT_SPLIT, int NOMEM = 0>
static hipError_t launch_trunk6_t(const TrunkArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trunk6_conv_kernel<RELU, RES, OUT_SPLIT, NOMEM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T6::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((trunk6_conv_kernel<RELU, RES, OUT_SPLIT, NOMEM>), dim3(grid), dim3(512), T6::LDS, s, a);
    return hipGetLastError();
}

// a.in / a.res mx6, a.out mx6 or (out_split) split-planar; ReLU layers and residual layers like launch_trunk_conv
hipError_t launch_trunk6_conv(TrunkArgs a, int out_split, hipStream_t s) {
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.out || !a.wpack || !a.bias) return hipErrorInvalidValue;
    if ((double)a.B * a.H * a.W * 256.0 + 4096.0 > 4294967295.0 || (double)a.zero_off + 4096.0 > 4294967295.0) return hipErrorInvalidValue;
    if ((a.relu != 0) == (a.res != nullptr)) return hipErrorInvalidValue;
    if (out_split && !a.res) return hipErrorInvalidValue;      // only the trunk's last layer leaves the format
    a.tiles_x = (a.W + T6::TWX - 1) / T6::TWX;
    a.tiles_y = (a.H + T6::TH - 1) / T6::TH;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    int npairs = std::max(1, trunk6_num_cus() / 16);
    npairs = std::min(npairs, per_xcd);
    const int rounds = (per_xcd + npairs - 1) / npairs;
    npairs = (per_xcd + rounds - 1) / rounds;
    const int grid = 16 * npairs;
    if (a.dbg && a.res && !out_split) {      // micro-benchmark ablations (tools/probe/trunk_ablate_b.py), residual layer
        switch (a.dbg) {
            case 1: return launch_trunk6_t<false, true, false, 1>(a, grid, s);
            case 4: return launch_trunk6_t<false, true, false, 4>(a, grid, s);
            case 5: return launch_trunk6_t<false, true, false, 5>(a, grid, s);
            case 16: return launch_trunk6_t<false, true, false, 16>(a, grid, s);
            case 21: return launch_trunk6_t<false, true, false, 21>(a, grid, s);      // MFMAs + fragment reads only
            case 29: return launch_trunk6_t<false, true, false, 29>(a, grid, s);      // MFMAs only
            case 23: return launch_trunk6_t<false, true, false, 23>(a, grid, s);      // fragment reads only
            case 30: return launch_trunk6_t<false, true, false, 30>(a, grid, s);      // tile DMA only
            case 27: return launch_trunk6_t<false, true, false, 27>(a, grid, s);      // epilogue memory only
            default: return launch_trunk6_t<false, true, false, 31>(a, grid, s);      // the empty pipeline
        }
    }
    if (a.res) return out_split ? launch_trunk6_t<false, true, true>(a, grid, s) : launch_trunk6_t<false, true, false>(a, grid, s);
    return launch_trunk6_t<true, false, false>(a, grid, s);
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
