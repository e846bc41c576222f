// FNet (reference layers.py:27-45) on the trunk's recipe: the 3x3 convolutions with 16..256 channels in bf16x3 arithmetic
// on SPLIT-PLANAR activations ([B][C/16 planes][H][W][64 B], conv_trunk.hip explains the record), staged global -> LDS by
// DMA, no LDS in the epilogue.  Differences from trunk_conv_kernel, all of them because FNet's launches are small and of
// many shapes (6..256 channels at 184x320 .. 23x40 pixels, 8 or 16 images per launch):
//  * a workgroup owns one 32-channel block of the outputs (Cout / 32 = 1..8 sibling workgroups share a tile, same XCD,
//    same time) and streams the weights plane by plane WITH the activations (18 KiB per 16 input channels, beside the
//    42 KiB halo tile, both double buffered): 256 -> 256 channels are 2.3 MiB of packed weights, nothing to keep resident;
//  * the batch is laid out as a MOSAIC for the tiling only: image i sits at cell (i / cm, i % cm) of a virtual picture with
//    gutters of 1 (2 for pooled layers: window parity) pixels between the cells.  Tiles of 8 x 64 pixels cut that picture,
//    not the single images, so an image of 23 x 40 or 46 x 80 pixels does not waste 40 % of every tile it touches.  The
//    gutters exist only in the DMA plan: a halo unit that falls into one reads the zero bytes behind the tensor (exactly the
//    zero padding the neighbouring images need), an output pixel in one goes to the dump area.  Memory stays per image;
//  * a wave owns two vertically adjacent rows of 32 pixels (M blocks), so MaxPooling2D (layers.py:31,33,35) is a max of
//    the wave's two accumulators plus one DPP exchange with the neighbouring lane, and the pooled record is stored from
//    the even lanes -- the unpooled tensor has no other reader and is never written;
//  * LeakyReLU(0.2) instead of ReLU, no residual; conv2d_13 writes fp32 NHWC for the taps-in-N kernel of conv2d_14.
#include "kernels.h"
#include "gather_math.h"

#include <algorithm>
#include <type_traits>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SR_AS1 __attribute__((address_space(1)))
#define SR_AS3 __attribute__((address_space(3)))

namespace {

struct FK {
    static constexpr int TH = 8, TWX = 64;
    static constexpr int HR = TH + 2, HC = TWX + 2;
    static constexpr int NPIX = HR * HC;                    // 660
    static constexpr int NUNIT = NPIX * 4;
    static constexpr int NINST = (NUNIT + 63) / 64;         // 42 DMA wave-instructions for the activations
    static constexpr int PER_WAVE = (NINST + 7) / 8;        // 6
    static constexpr int A_BYTES = NINST * 1024;            // 43,008
    static constexpr int W_BYTES = 9 * 2 * 1024;            // 18,432: [tap][hi, lo][lane] x 16 B of one plane, one 32-channel block
    static constexpr int W_PER_WAVE = 3;                    // 18 DMA wave-instructions, waves 0, 1 a third one
    static constexpr int BUF = W_BYTES + A_BYTES;           // 61,440
    static constexpr int LDS = 2 * BUF;                     // 122,880 (+ 128 bytes of biases behind it)
    static constexpr int LDS_ALL = LDS + 128;
};

__device__ __forceinline__ float bf_even(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_odd(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// max(v, 0.2 v) as one multiply and one v_max (fmaxf() puts a canonicalising v_max in front)
__device__ __forceinline__ float lrelu(float v) { return __builtin_amdgcn_fmed3f(v, 0.2f * v, __builtin_inff()); }
__device__ __forceinline__ float max2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
// the value of the neighbouring lane (r ^ 1)
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

}  // namespace

// NPL: input planes (Cin / 16).  POOL: store the 2x2 max pooling of the activated output.  OUT_F32: fp32 NHWC output.
template <int NPL, bool POOL, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void fnet_conv_kernel(const FConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wj = wave >> 1, wi = wave & 1;            // the wave's pixels: tile rows 2 wj, 2 wj + 1, columns 32 wi + r
    const int H = a.H, W = a.W;

    // workgroup -> (XCD, output block, group): ids with equal (id & 7) share an L2; the blocks of a group take the same tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = slot % a.ncb, grp = slot / a.ncb;
    const int ngroups = (int)(gridDim.x >> 3) / a.ncb;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int per_xcd = (ntiles + 7) / 8;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int t = xcd * per_xcd + grp;
    if (t >= t_end) return;

    const unsigned plane_bytes = (unsigned)(H * W * 64);
    const unsigned img_bytes = (unsigned)NPL * plane_bytes;
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const unsigned oplane = (unsigned)(Ho * Wo * 64);                 // split-planar output plane
    const unsigned oimg = OUT_F32 ? (unsigned)(H * W) * (unsigned)(a.ncb * 128) : (unsigned)(2 * a.ncb) * oplane;

    // ---- per-lane plan of the activation DMA: 42 instructions of 1 KiB, wave w issues k = w + 8 i (waves >= 2 repeat their
    // fifth as a sixth).  Unit u = 64 k + lane = LDS unit (u & 3) of halo pixel u >> 2; its source unit is XOR-swizzled by
    // (pixel >> 2) & 3 = (lane >> 4) & 3 for every k.
    int kq[FK::PER_WAVE];
    int pyx[FK::PER_WAVE];         // halo column | halo row << 8; row 2^20: no such unit
#pragma unroll
    for (int i = 0; i < FK::PER_WAVE; ++i) {
        kq[i] = wave + 8 * i < FK::NINST ? wave + 8 * i : wave + 8 * (i - 1);
        const int u = kq[i] * 64 + lane;
        const int p = u >> 2;
        const int py = p / FK::HC, px = p - py * FK::HC;
        pyx[i] = u < FK::NUNIT ? (px | (py << 8)) : (1 << 28);
        asm volatile("" : "+v"(pyx[i]));
    }
    const unsigned sl16 = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const unsigned zoff = a.zero_off + (lane & 3) * 16;
    const unsigned dump = a.dump_off + 256 + lane * 32;

    // mosaic coordinate -> (cell, offset in cell); exact for coordinates below 2^16 (launcher checks)
    auto cell_of = [](int g, float inv, int cell, int& rem) {
        const int c = (int)(((float)g + 0.5f) * inv);
        rem = g - c * cell;
        return c;
    };
    struct Tile {
        unsigned off[FK::PER_WAVE];   // source offset of this lane's unit of DMA instruction i, plane 0
        unsigned okm;                 // bit i: that unit lies inside an image
        unsigned eoff[2];             // where this lane's output record of M block mb (plane / channel group 0) goes, or `dump`
    };
    auto locate = [&](int tt) {
        Tile q;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int y0 = ty * FK::TH - 1, x0 = tx * FK::TWX - 1;
        q.okm = 0;
#pragma unroll
        for (int i = 0; i < FK::PER_WAVE; ++i) {
            const int gy = y0 + (pyx[i] >> 8), gx = x0 + (pyx[i] & 255);
            int ry, rx;
            const int cy = cell_of(gy, a.inv_ch, a.cell_h, ry), cx = cell_of(gx, a.inv_cw, a.cell_w, rx);
            const int img = cy * a.cm + cx;
            const bool ok = ((unsigned)gy < (unsigned)a.MH) & ((unsigned)gx < (unsigned)a.MW) & (ry < H) & (rx < W) & (img < a.B);
            q.okm |= ok ? (1u << i) : 0u;
            q.off[i] = (unsigned)img * img_bytes + (unsigned)((ry * W + rx) * 64) + sl16;
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int gy = y0 + 1 + 2 * wj + mb, gx = x0 + 1 + 32 * wi + r;
            int ry, rx;
            const int cy = cell_of(gy, a.inv_ch, a.cell_h, ry), cx = cell_of(gx, a.inv_cw, a.cell_w, rx);
            const int img = cy * a.cm + cx;
            bool ok = (gy < a.MH) & (gx < a.MW) & (ry < H) & (rx < W) & (img < a.B);
            unsigned o;
            if (POOL) {
                ok = ok & ((r & 1) == 0) & (mb == 0);
                o = (unsigned)img * oimg + (unsigned)(2 * cb) * oplane + (unsigned)(((ry >> 1) * Wo + (rx >> 1)) * 64) + 32u * h;
            } else if (OUT_F32) {
                o = (unsigned)img * oimg + (unsigned)(ry * W + rx) * (unsigned)(a.ncb * 128) + (unsigned)(cb * 128) + 16u * h;
            } else {
                o = (unsigned)img * oimg + (unsigned)(2 * cb) * oplane + (unsigned)((ry * W + rx) * 64) + 32u * h;
            }
            q.eoff[mb] = ok ? o : dump;
        }
        return q;
    };
    const unsigned char* const wsrc = a.wpack + (size_t)cb * NPL * FK::W_BYTES + lane * 16;
    auto issue = [&](const Tile& q, int c, int buf) {
        unsigned char* const wdst = smem + buf * FK::BUF;
        unsigned char* const adst = wdst + FK::W_BYTES;
        const unsigned offc = (unsigned)c * plane_bytes;
#pragma unroll
        for (int i = 0; i < FK::PER_WAVE; ++i) {
            const unsigned off = ((q.okm >> i) & 1u) ? q.off[i] + offc : zoff;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.in + off), (SR_AS3 void*)(adst + kq[i] * 1024), 16, 0, 0);
        }
        const unsigned char* ws = wsrc + (size_t)c * FK::W_BYTES;
#pragma unroll
        for (int i = 0; i < FK::W_PER_WAVE; ++i) {
            const int k = wave + 8 * i < 18 ? wave + 8 * i : wave + 8 * (i - 1);
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(ws + k * 1024), (SR_AS3 void*)(wdst + k * 1024), 16, 0, 0);
        }
    };
    Tile cur = locate(t);
    issue(cur, 0, 0);

    // the block's 32 biases wait in LDS (behind the two buffers) and are read at the start of every tile: sixteen registers a
    // lane would otherwise hold for the whole launch are what the widest instantiations lack (a spill would put scratch
    // loads into the counted vmcnt queue)
    float* const bias_lds = reinterpret_cast<float*>(smem + FK::LDS);
    if (tid < 32) bias_lds[tid] = a.bias[32 * cb + tid];

    // B operand = pixels: lane (r, h) reads unit 2h (hi) and 2h+1 (lo) of halo pixel (2 wj + mb + ky, 32 wi + r + kx)
    auto a_addr = [&](int ky, int kx, int mb) {
        const int p = (2 * wj + mb + ky) * FK::HC + 32 * wi + r + kx;
        return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4);
    };

    // One piece of a finished tile's epilogue = 8 channels of one record = 4 packed hi / lo pairs.
    //   plain: piece (mb, pl): channels 8 pl .. 8 pl + 7 of M block mb;  POOL: piece pl of the pooled block
    constexpr int NP = POOL ? 2 : 4;                            // pieces per tile
    constexpr int PPS = (NP + NPL - 1) / NPL;                   // pieces per plane step, in the first NP / PPS steps
    constexpr int EP_STEPS = NP / PPS;
    f32x16 acc[2], fin[2];
    unsigned done_eoff[2] = {dump, dump};
#pragma unroll
    for (int i = 0; i < 16; ++i) fin[0][i] = fin[1][i] = 0.f;

    struct Piece {
        u32x4 oh, ol;
        f32x4 f0, f1;
        unsigned off;
    };
    auto ep_value = [&](int pc, int i) {          // activated value i (0..7) of piece pc
        if (POOL) {
            const float m = max2(fin[0][8 * pc + i], fin[1][8 * pc + i]);
            return lrelu(max2(m, lane_xor1(m)));
        }
        return lrelu(fin[pc >> 1][8 * (pc & 1) + i]);
    };
    auto ep_pair = [&](Piece& p, int pc, int k) {
        const float v0 = ep_value(pc, 2 * k), v1 = ep_value(pc, 2 * k + 1);
        if (OUT_F32) {
            if (k < 2) {
                p.f0[2 * k] = v0;
                p.f0[2 * k + 1] = v1;
            } else {
                p.f1[2 * k - 4] = v0;
                p.f1[2 * k - 3] = v1;
            }
        } else {
            p.oh[k] = pack_bf16(v0, v1);
            p.ol[k] = pack_bf16(v0 - bf_even(p.oh[k]), v1 - bf_odd(p.oh[k]));
        }
    };
    auto ep_store = [&](Piece& p, int pc) {
        if (OUT_F32) {          // channels 8 (2 pl) + 4 h .. and 8 (2 pl + 1) + 4 h .. of the block: 32 bytes apart
            const unsigned o = done_eoff[pc >> 1] + 64u * (pc & 1);
            *reinterpret_cast<f32x4*>(a.out + o) = p.f0;
            *reinterpret_cast<f32x4*>(a.out + o + 32) = p.f1;
        } else {
            const int mb = POOL ? 0 : pc >> 1, pl = POOL ? pc : pc & 1;
            const unsigned o = done_eoff[mb] + (pl && done_eoff[mb] != dump ? oplane : 0u);
            *reinterpret_cast<u32x4*>(a.out + o) = p.oh;
            *reinterpret_cast<u32x4*>(a.out + o + 16) = p.ol;
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first plane's tile and weights
    __syncthreads();                                   // the biases
    const int npg = ngroups;
    int q = 0;
    for (;;) {
        const int tn = t + npg;
        const bool have_next = tn < t_end;
        const Tile nxt = locate(have_next ? tn : t);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][4 * g + j] = acc[1][4 * g + j] = b4[j];
        }

        // One plane step.  CI: its index among the (up to four) steps of a group -- a tile's NPL steps run as NPL / 4 groups,
        // the first one unrolled with the finished tile's epilogue pieces, the others as a run-time loop (fully unrolled, 16
        // planes keep 16 sets of hoisted addresses alive and spill; a spill is a scratch load in the hand-counted vmcnt
        // queue).  EPI: the group that carries the epilogue.  c: the plane (run time for NPL > 4).
        // LAST: 0 = another plane of this tile follows, 1 = the tile's last plane, 2 = ask c.
        auto plane_step = [&](auto ci_, auto epi_, auto last_, int c) {
            constexpr int CI = decltype(ci_)::value;
            constexpr bool EPI = decltype(epi_)::value;
            constexpr int LAST = decltype(last_)::value;
            constexpr int NEP = EPI && CI < EP_STEPS ? PPS : 0;       // epilogue pieces of the finished tile under this step
            // stores younger than this step's DMA = those of the step before it: step CI - 1 of this group, or (CI = 0) the
            // last step of the previous tile when NPL <= 4, none when a longer tile ended with a group without epilogue; the
            // first step of such a group waits for everything (the epilogue group's stores may still be in flight)
            constexpr int PREVI = EPI ? (CI + NPL - 1) % NPL : -1;
            constexpr int YOUNGER = EPI ? (CI > 0 ? 2 * (CI - 1 < EP_STEPS ? PPS : 0) : (NPL <= 4 ? 2 * (PREVI < EP_STEPS ? PPS : 0) : 0)) : 0;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            __builtin_amdgcn_s_barrier();           // every wave's part landed; every wave is done reading the other buffer
            asm volatile("" ::: "memory");
            if (LAST == 0) {
                issue(cur, c + 1, (q + 1) & 1);
            } else if (LAST == 1) {
                issue(nxt, 0, (q + 1) & 1);         // past the last tile: re-reads this tile's first plane, harmless
            } else {                                // wave-uniform choice without a branch: the step stays one scheduling region
                const bool last = c + 1 >= NPL;
                Tile sel;
#pragma unroll
                for (int i = 0; i < FK::PER_WAVE; ++i) sel.off[i] = last ? nxt.off[i] : cur.off[i];
                sel.okm = last ? nxt.okm : cur.okm;
                issue(sel, last ? 0 : c + 1, (q + 1) & 1);
            }
            const unsigned char* bw = smem + (q & 1) * FK::BUF + lane * 16;
            const unsigned char* at = smem + (q & 1) * FK::BUF + FK::W_BYTES;
            ++q;
            Piece pcs[NEP ? NEP : 1];
            u32x4 fa[2][2][2], fb[2][2];            // [buffer][M block][hi, lo], [buffer][hi, lo]
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
            load_tap(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = tap & 1;
                if (tap + 1 < 9) load_tap(tap + 1, s ^ 1);
                const bf16x8 wh = __builtin_bit_cast(bf16x8, fb[s][0]), wlo = __builtin_bit_cast(bf16x8, fb[s][1]);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, fa[s][mb][0]), xl = __builtin_bit_cast(bf16x8, fa[s][mb][1]);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[mb], 0, 0, 0);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh, acc[mb], 0, 0, 0);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[mb], 0, 0, 0);
                }
                // the epilogue pieces, spread over taps 1..6: pair n of the step's 4 NEP pairs under tap 1 + 6 n / (4 NEP); stores under tap 7
                if constexpr (NEP > 0) {
#pragma unroll
                    for (int n = 0; n < 4 * NEP; ++n)
                        if (1 + (6 * n) / (4 * NEP) == tap) ep_pair(pcs[n >> 2], CI * PPS + (n >> 2), n & 3);
                    if (tap == 7) {
#pragma unroll
                        for (int e = 0; e < NEP; ++e) ep_store(pcs[e], CI * PPS + e);
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (tap == 0) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                    if (tap == 0) __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x006, NEP > 1 ? 4 + 3 * NEP : 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        if constexpr (NPL <= 4) {     // one group, everything static
            static_for<NPL>([&](auto cc) {
                constexpr int C = decltype(cc)::value;
                plane_step(cc, true_type{}, integral_constant<int, (C + 1 == NPL) ? 1 : 0>{}, C);
            });
        } else {
            plane_step(integral_constant<int, 0>{}, true_type{}, integral_constant<int, 0>{}, 0);
            plane_step(integral_constant<int, 1>{}, true_type{}, integral_constant<int, 0>{}, 1);
            plane_step(integral_constant<int, 2>{}, true_type{}, integral_constant<int, 0>{}, 2);
            plane_step(integral_constant<int, 3>{}, true_type{}, integral_constant<int, 0>{}, 3);
#pragma unroll 1
            for (int c0 = 4; c0 < NPL; c0 += 4) {
                plane_step(integral_constant<int, 0>{}, false_type{}, integral_constant<int, 0>{}, c0);
                plane_step(integral_constant<int, 1>{}, false_type{}, integral_constant<int, 0>{}, c0 + 1);
                plane_step(integral_constant<int, 2>{}, false_type{}, integral_constant<int, 0>{}, c0 + 2);
                plane_step(integral_constant<int, 3>{}, false_type{}, integral_constant<int, 2>{}, c0 + 3);
            }
        }
        fin[0] = acc[0];
        fin[1] = acc[1];
        done_eoff[0] = cur.eoff[0];
        done_eoff[1] = cur.eoff[1];
        if (!have_next) break;
        t = tn;
        cur = nxt;
    }
    // the last tile's epilogue has no MFMAs to hide under
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
        Piece p;
#pragma unroll
        for (int k = 0; k < 4; ++k) ep_pair(p, pc, k);
        ep_store(p, pc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing DMA (into LDS) must not outlive the workgroup
}

// ------------------------------------------------------------------ element-wise neighbours on split-planar data
// concat [LR_t, LR_t-1] (layers.py:29), zero padded to one 16-channel plane: record = hi / lo units of channels
// {0..3, 8..11} (h = 0) and {4..7, 12..15} (h = 1); thread = (pixel, h)
__global__ void fnet_input_split_kernel(const float* __restrict__ a, const float* __restrict__ b, unsigned char* __restrict__ out, unsigned npix) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned p = idx >> 1, hh = idx & 1;
    if (p >= npix) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * hh + e;               // channels 8.. of the unit are zero
        v[e] = c < 3 ? a[p * 3 + c] : (c < 6 ? b[p * 3 + c - 3] : 0.f);
    }
    u32x4 hi = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), 0u, 0u};
    u32x4 lo = {pack_bf16(v[0] - bf_even(hi[0]), v[1] - bf_odd(hi[0])), pack_bf16(v[2] - bf_even(hi[1]), v[3] - bf_odd(hi[1])), 0u, 0u};
    unsigned char* dst = out + (size_t)p * 64 + 32 * hh;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 16) = lo;
}

hipError_t launch_fnet_input_split(const float* lr_t, const float* lr_tm1, void* out, int B, int H, int W, hipStream_t s) {
    const double n = (double)B * H * W;
    if (n * 2 >= 4294967295.0) return hipErrorInvalidValue;
    const unsigned npix = (unsigned)n;
    hipLaunchKernelGGL(fnet_input_split_kernel, dim3((2 * npix + 255) / 256), dim3(256), 0, s, lr_t, lr_tm1, (unsigned char*)out, npix);
    return hipGetLastError();
}

// TF1-legacy bilinear x2 (layers.py:37,39,41) split-planar -> split-planar: thread = (output pixel, h) of one plane of one image;
// value = hi + lo (exact in fp32), lerp, split again
__global__ void resize2x_split_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, unsigned nplanes) {
    const int Wo = 2 * W, Ho = 2 * H;
    const unsigned per_plane = (unsigned)(Ho * Wo) * 2u;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned pl = blockIdx.y;
    if (idx >= per_plane || pl >= nplanes) return;
    const unsigned hh = idx & 1, p = idx >> 1;
    const int oy = (int)(p / (unsigned)Wo), ox = (int)(p - (unsigned)oy * (unsigned)Wo);
    const int y0 = oy >> 1, x0 = ox >> 1;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ty = (oy & 1) * 0.5f, tx = (ox & 1) * 0.5f;
    const unsigned char* base = in + (size_t)pl * H * W * 64 + 32 * hh;
    auto rd = [&](int y, int x, float* v) {
        const u32x4 hi = *reinterpret_cast<const u32x4*>(base + (size_t)(y * W + x) * 64);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(base + (size_t)(y * W + x) * 64 + 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = bf_even(hi[k]) + bf_even(lo[k]);
            v[2 * k + 1] = bf_odd(hi[k]) + bf_odd(lo[k]);
        }
    };
    float tl[8], tr[8], bl[8], br[8];
    rd(y0, x0, tl);
    rd(y0, x1, tr);
    rd(y1, x0, bl);
    rd(y1, x1, br);
    u32x4 oh, ol;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v0 = lerp2(tl[2 * k], tr[2 * k], bl[2 * k], br[2 * k], tx, ty);
        const float v1 = lerp2(tl[2 * k + 1], tr[2 * k + 1], bl[2 * k + 1], br[2 * k + 1], tx, ty);
        oh[k] = pack_bf16(v0, v1);
        ol[k] = pack_bf16(v0 - bf_even(oh[k]), v1 - bf_odd(oh[k]));
    }
    unsigned char* dst = out + (size_t)pl * Ho * Wo * 64 + (size_t)p * 64 + 32 * hh;
    *reinterpret_cast<u32x4*>(dst) = oh;
    *reinterpret_cast<u32x4*>(dst + 16) = ol;
}

hipError_t launch_resize2x_split(const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
    if (C % 16 || (double)H * W * 8 >= 4294967295.0 || (double)B * (C / 16) > 65535.0) return hipErrorInvalidValue;
    const unsigned per_plane = (unsigned)(4 * H * W) * 2u;
    hipLaunchKernelGGL(resize2x_split_kernel, dim3((per_plane + 255) / 256, (unsigned)(B * (C / 16))), dim3(256), 0, s,
                       (const unsigned char*)in, (unsigned char*)out, H, W, (unsigned)(B * (C / 16)));
    return hipGetLastError();
}

// ------------------------------------------------------------------ launcher
static int fnet_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int NPL, bool POOL, bool OUT_F32>
static hipError_t launch_fnet_t(const FConvArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fnet_conv_kernel<NPL, POOL, OUT_F32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, FK::LDS_ALL);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((fnet_conv_kernel<NPL, POOL, OUT_F32>), dim3(grid), dim3(512), FK::LDS_ALL, s, a);
    return hipGetLastError();
}

size_t fnet_weight_bytes(int cin, int cout) { return (size_t)((cout + 31) / 32) * ((cin + 15) / 16) * FK::W_BYTES; }

// The mosaic for B images of H x W with gutter g: cm images per mosaic row.  Picks the cm whose tiled area is smallest.
static void choose_mosaic(int B, int H, int W, int g, int& cm, int& tiles_x, int& tiles_y) {
    double best = 1e300;
    cm = 1;
    for (int c = 1; c <= B; ++c) {
        const int rm = (B + c - 1) / c;
        const int MW = c * (W + g) - g, MH = rm * (H + g) - g;
        if (MW >= 65536 || MH >= 65536) continue;
        const int tx = (MW + FK::TWX - 1) / FK::TWX, ty = (MH + FK::TH - 1) / FK::TH;
        const double area = (double)tx * ty;
        if (area < best) {
            best = area;
            cm = c;
            tiles_x = tx;
            tiles_y = ty;
        }
    }
}

hipError_t launch_fnet_conv(FConvArgs a, hipStream_t s) {
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.out || !a.wpack || !a.bias || a.ncb < 1 || a.ncb > 8) return hipErrorInvalidValue;
    if (a.pool && ((a.H & 1) || (a.W & 1) || a.out_f32)) return hipErrorInvalidValue;
    // 32-bit offsets: both tensors with their 4 KiB tails
    const double in_bytes = (double)a.B * a.npl * a.H * a.W * 64.0, out_bytes = (double)a.B * a.ncb * 2 * a.H * a.W * 64.0;
    if (in_bytes + 4096.0 > 4294967295.0 || out_bytes + 4096.0 > 4294967295.0 || (double)a.zero_off + 4096.0 > 4294967295.0 ||
        (double)a.dump_off + 4096.0 > 4294967295.0)
        return hipErrorInvalidValue;
    const int g = a.pool ? 2 : 1;
    choose_mosaic(a.B, a.H, a.W, g, a.cm, a.tiles_x, a.tiles_y);
    const int rm = (a.B + a.cm - 1) / a.cm;
    a.cell_w = a.W + g;
    a.cell_h = a.H + g;
    a.MW = a.cm * a.cell_w - g;
    a.MH = rm * a.cell_h - g;
    if (a.MW >= 65536 || a.MH >= 65536) return hipErrorInvalidValue;
    a.inv_cw = 1.0f / (float)a.cell_w;
    a.inv_ch = 1.0f / (float)a.cell_h;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int per_xcd = (ntiles + 7) / 8;
    int ngroups = std::max(1, fnet_num_cus() / 8 / a.ncb);
    ngroups = std::min(ngroups, per_xcd);
    const int rounds = (per_xcd + ngroups - 1) / ngroups;
    ngroups = (per_xcd + rounds - 1) / rounds;
    const int grid = 8 * a.ncb * ngroups;
    const int key = a.npl * 4 + (a.pool ? 1 : 0) + (a.out_f32 ? 2 : 0);
    switch (key) {
        case 1 * 4 + 0: return launch_fnet_t<1, false, false>(a, grid, s);
        case 2 * 4 + 0: return launch_fnet_t<2, false, false>(a, grid, s);
        case 2 * 4 + 1: return launch_fnet_t<2, true, false>(a, grid, s);
        case 4 * 4 + 0: return launch_fnet_t<4, false, false>(a, grid, s);
        case 4 * 4 + 1: return launch_fnet_t<4, true, false>(a, grid, s);
        case 4 * 4 + 2: return launch_fnet_t<4, false, true>(a, grid, s);
        case 8 * 4 + 0: return launch_fnet_t<8, false, false>(a, grid, s);
        case 8 * 4 + 1: return launch_fnet_t<8, true, false>(a, grid, s);
        case 16 * 4 + 0: return launch_fnet_t<16, false, false>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sr
