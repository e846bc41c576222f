// SRNet trunk: the 64 -> 64 3x3 convolutions of reference layers.py:58-62,67-70 (head + 10 residual blocks = 21
// launches per frame, two thirds of the generator's FLOPs) in bf16x3 arithmetic on SPLIT-PLANAR activations.
//
// Data layout ("split planar"): [B][4 planes][H][W][64 B].  A plane holds 16 channels of every pixel as a 64-byte
// record of four 16-byte units
//     unit 2h   : bf16 hi parts of the 8 channels  16*plane + {4h .. 4h+3, 8+4h .. 8+4h+3}
//     unit 2h+1 : bf16 lo parts of the same channels            (h = 0, 1;  value = hi + lo)
// i.e. the fp32 activation is stored already split into the two bf16 terms the three-MFMA product needs
// (w_hi*x_lo + w_lo*x_hi + w_hi*x_hi), 4 bytes per element like the fp32 tensor it replaces.  The odd channel order
// is the accumulator layout of v_mfma_f32_32x32x16_bf16: lane (pixel, h) owns exactly the channels of unit 2h / 2h+1,
// so the epilogue stores a lane's 8 + 8 values as two 16-byte vectors with no transpose, and the same 16 bytes
// are what a lane of the next layer's B operand reads (k index (h, e) <-> channel; the weight packer follows it).
//
// Why: with fp32 activations the staging of a chunk was load -> VALU split -> LDS store through registers, 2/3 of a
// tile's time (profiles/r1_sq_counters.txt); here staging is a plain copy and goes global -> LDS directly
// (global_load_lds_dwordx4, no registers, no VALU, no ds_write), and the epilogue needs no LDS at all.
//
// Decomposition: persistent 512-thread workgroups, one per CU.  A workgroup owns ONE half of the output channels
// (32 of 64) for the whole launch and keeps that half's packed weights for all 64 input channels resident in LDS
// (72 KiB, fetched once), so per tile only activations move: the 10 x 66 pixel halo tile of one 16-channel plane
// (42 KiB) is double buffered, the next plane (or the next tile's first plane) landing while the current one
// feeds the MFMAs; one workgroup barrier per plane.  Tile = 8 rows x 64 columns, wave w computes row w as two
// 32-pixel M blocks x 32 output channels.  The two halves of a tile run on sibling workgroups of the same XCD at
// the same time, so the second read of the tile's activations is an L2 hit.
// LDS image of a halo pixel = its 64-byte record with the 16-byte units XOR-swizzled by (pixel >> 2) & 3 (applied
// to the SOURCE address, the DMA writes LDS linearly): the 32 lanes of a ds_read_b128 fragment read then fall on 16
// distinct 16-byte bank columns per 16-lane group -- conflict free without padding.
#include "kernels.h"

#include <algorithm>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SR_AS1 __attribute__((address_space(1)))
#define SR_AS3 __attribute__((address_space(3)))

namespace {

struct TK {
    static constexpr int TH = 8, TWX = 64;                  // output tile: one row per wave, two 32-pixel M blocks
    static constexpr int HR = TH + 2, HC = TWX + 2;         // halo tile
    static constexpr int NPIX = HR * HC;                    // 660
    static constexpr int NUNIT = NPIX * 4;                  // 16-byte units per plane tile
    static constexpr int NINST = (NUNIT + 63) / 64;         // 42 DMA wave-instructions (1 KiB each)
    static constexpr int PER_WAVE = (NINST + 7) / 8;        // 6 (waves 0, 1) or 5
    static constexpr int A_BYTES = NINST * 1024;            // 43,008 per buffer
    static constexpr int W_BYTES = 4 * 9 * 2 * 1024;        // 73,728: [chunk][tap][hi,lo][lane] x 16 B
    static constexpr int LDS = W_BYTES + 2 * A_BYTES;       // 159,744 <= 163,840
};

// the two bf16 of a packed dword as floats: even element (low half), odd element (high half)
__device__ __forceinline__ float bf_even(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_odd(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }

// two floats -> two bf16 (round to nearest even) packed low / high
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

}  // namespace

__global__ __launch_bounds__(512, 2) void trunk_conv_kernel(const TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wl = smem;                 // resident weights of this workgroup's channel half
    unsigned char* const abuf = smem + TK::W_BYTES; // two plane tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;

    // workgroup -> (XCD, channel half, pair): workgroup ids go round-robin over the 8 XCDs, so ids with equal
    // (id & 7) share an L2; the two halves of a pair take the same tiles at the same time
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int half = slot & 1, pair = slot >> 1;
    const int npairs = (int)(gridDim.x >> 4);
    const int per_img = a.tiles_x * a.tiles_y;
    const int ntiles = per_img * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int t = xcd * per_xcd + pair;
    if (t >= t_end) return;

    const size_t plane_bytes = (size_t)H * W * 64, img_bytes = plane_bytes * 4;   // the launcher keeps B * img_bytes < 4 GiB

    // ---- resident weights: 72 DMA instructions, 9 per wave
    {
        const unsigned char* wsrc = a.wpack + (size_t)half * TK::W_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(wsrc + k * 1024), (SR_AS3 void*)(wl + k * 1024), 16, 0, 0);
        }
    }

    // ---- per-lane plan of the plane-tile DMA: instruction k = wave + 8 i moves units k*64 + lane
    int goff[TK::PER_WAVE];        // byte offset of the unit's source from the halo origin (row y0-1, column x0-1)
    int pyx[TK::PER_WAVE];         // halo column | halo row << 8; row 2^20: no such unit
#pragma unroll
    for (int i = 0; i < TK::PER_WAVE; ++i) {
        const int u = (wave + 8 * i) * 64 + lane;
        const int p = u >> 2, sl = (u & 3) ^ ((p >> 2) & 3);     // source unit of LDS unit (u & 3) of pixel p
        const int py = p / TK::HC, px = p - py * TK::HC;
        const bool exists = u < TK::NUNIT;
        goff[i] = (py * W + px) * 64 + sl * 16;
        pyx[i] = exists ? (px | (py << 8)) : (1 << 28);
        asm volatile("" : "+v"(goff[i]), "+v"(pyx[i]));          // keep them in registers: recomputing costs a division per DMA
    }
    const unsigned zoff = a.zero_off + (lane & 3) * 16;          // 64 zero bytes behind the tensor: halo outside the image
    // addresses are a.in + 32-bit offset (the tensor is < 4 GiB): SGPR base + one VGPR per DMA
    auto issue = [&](int tt, int c, int buf) {
        const int b = tt / per_img;
        const int rem = tt - b * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int y0 = ty * TK::TH - 1, x0 = tx * TK::TWX - 1;
        const unsigned tile_off = (unsigned)(b * 4 + c) * (unsigned)plane_bytes + (unsigned)((y0 * W + x0) * 64);   // wraps for y0 = -1: only valid units use it
        unsigned char* dst = abuf + buf * TK::A_BYTES;
#pragma unroll
        for (int i = 0; i < TK::PER_WAVE; ++i) {
            const int k = wave + 8 * i;
            if (k < TK::NINST && !(a.dbg & 1)) {   // wave-uniform
                const unsigned gy = (unsigned)(y0 + (pyx[i] >> 8)), gx = (unsigned)(x0 + (pyx[i] & 255));
                const bool ok = (gy < (unsigned)H) & (gx < (unsigned)W);   // a unit that does not exist fails the row test
                const unsigned off = ok ? tile_off + (unsigned)goff[i] : zoff;
                __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.in + off), (SR_AS3 void*)(dst + k * 1024), 16, 0, 0);
            }
        }
    };
    issue(t, 0, 0);

    // ---- fragment addresses (bytes from the start of a plane tile / of the weights)
    // B operand = pixels: lane (r, h) reads unit 2h (hi) and 2h+1 (lo) of halo pixel (wave + ky, 32 mb + r + kx)
    auto a_addr = [&](int ky, int kx, int mb) {
        const int p = (wave + ky) * TK::HC + mb * 32 + r + kx;
        return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4);
    };
    const unsigned char* const bl = wl + lane * 16;

    const float slope = a.relu ? 0.f : 1.f;
    int q = 0;            // running count of (tile, plane) steps: buffer = q & 1
    int nst = 0;          // vector stores this wave issued in the previous tile's epilogue, after its last DMA
    for (;;) {
        const int tn = t + npairs;
        const bool have_next = tn < t_end;
        f32x16 acc[2];
        {   // the accumulators start from the bias: lane (pixel r, h) owns channels 32 half + 8 g + 4 h + (0..3)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 32 * half + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][4 * g + j] = acc[1][4 * g + j] = bv[j];
            }
        }
#pragma unroll 1
        for (int c = 0; c < 4; ++c, ++q) {
            // this wave's DMA of step q has landed.  vmcnt counts in issue order, so after a full epilogue (8 stores
            // issued behind the DMA) the stores may stay in flight; any other case drains everything
            if (c == 0 && nst == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // every wave's part landed; every wave is done reading the other buffer
            asm volatile("" ::: "memory");
            if (c < 3) issue(t, c + 1, (q + 1) & 1);
            else if (have_next) issue(tn, 0, (q + 1) & 1);

            const unsigned char* at = abuf + (q & 1) * TK::A_BYTES;
            const unsigned char* bw = bl + c * (9 * 2 * 1024);
            if (a.dbg & 2) continue;
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
                if (tap + 1 < 9) {   // one LDS read behind each MFMA
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: activation, residual, split, two 16-byte stores per (M block, plane)
        {
            const int b = t / per_img;
            const int rem = t - b * per_img;
            const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
            const int y = ty * TK::TH + wave;
            // a store instruction is issued when at least one of its lanes is inside the image (wave-uniform count)
            nst = (y < H && !(a.dbg & 4)) ? (tx * TK::TWX + 32 < W ? 8 : 4) : 0;
            const size_t row_off = (size_t)b * img_bytes + (size_t)(2 * half) * plane_bytes + (size_t)y * W * 64 + 32 * h;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int x = tx * TK::TWX + mb * 32 + r;
                const bool ok = y < H && x < W;
                const size_t off = row_off + (size_t)x * 64;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float tv = acc[mb][8 * pl + e];      // g = 2 pl + (e >> 2), j = e & 3
                        v[e] = fmaxf(tv, slope * tv);
                    }
                    const size_t o = off + (size_t)pl * plane_bytes;
                    if (a.res) {
                        u32x4 rh = {0u, 0u, 0u, 0u}, rl = {0u, 0u, 0u, 0u};
                        if (ok && !(a.dbg & 4)) {
                            rh = *reinterpret_cast<const u32x4*>(a.res + o);
                            rl = *reinterpret_cast<const u32x4*>(a.res + o + 16);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[2 * k] += bf_even(rh[k]) + bf_even(rl[k]);
                            v[2 * k + 1] += bf_odd(rh[k]) + bf_odd(rl[k]);
                        }
                    }
                    u32x4 oh, ol;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        oh[k] = pack_bf16(v[2 * k], v[2 * k + 1]);
                        ol[k] = pack_bf16(v[2 * k] - bf_even(oh[k]), v[2 * k + 1] - bf_odd(oh[k]));
                    }
                    if (ok && (!(a.dbg & 4) || oh[0] == 0x12345678u)) {
                        *reinterpret_cast<u32x4*>(a.out + o) = oh;
                        *reinterpret_cast<u32x4*>(a.out + o + 16) = ol;
                    }
                }
            }
        }
        if (!have_next) break;
        t = tn;
    }
}

static int trunk_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

size_t trunk_weight_bytes() { return (size_t)2 * TK::W_BYTES; }

hipError_t launch_trunk_conv(TrunkArgs a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trunk_conv_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TK::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.out || !a.wpack || !a.bias) return hipErrorInvalidValue;
    if ((double)a.B * a.H * a.W * 256.0 + 64.0 > 4294967295.0 || (double)a.zero_off + 64.0 > 4294967295.0) return hipErrorInvalidValue;
    a.tiles_x = (a.W + TK::TWX - 1) / TK::TWX;
    a.tiles_y = (a.H + TK::TH - 1) / TK::TH;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    // one workgroup per CU, in (XCD, half, pair) order: 16 workgroups per pair index; no more pairs than an XCD has tiles
    const int per_xcd = (ntiles + 7) / 8;
    int npairs = std::max(1, trunk_num_cus() / 16);
    npairs = std::min(npairs, per_xcd);
    // even out the rounds: every pair the same number of tiles (+-1)
    const int rounds = (per_xcd + npairs - 1) / npairs;
    npairs = (per_xcd + rounds - 1) / rounds;
    hipLaunchKernelGGL(trunk_conv_kernel, dim3(16 * npairs), dim3(512), TK::LDS, s, a);
    return hipGetLastError();
}

}  // namespace sr
