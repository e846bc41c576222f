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
#include "mx6_format.h"

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

// RELU: ReLU after the bias; RES: add a.res after it.  NOMEM (ablation builds of the micro-benchmark only): 1 no tile
// DMA, 2 no MFMAs, 4 no epilogue loads / stores.
//
// Software pipeline inside a wave: the epilogue of tile n (activation, residual, split into hi / lo, stores) is cut into
// four pieces -- (M block, plane) -- and piece c runs inside the MFMA stream of plane c of tile n+1, on a copy of the
// finished accumulators.  The two waves of a SIMD run this same program in lock step (one barrier per plane), so work
// that is not interleaved with a wave's own MFMAs is not covered by its partner either: as separate phases the
// epilogue VALU (about 750 instructions per wave and tile) and the DMA address arithmetic took as long as the MFMAs.
// Everything in a plane's body is branch free so that it stays one scheduling region: lanes outside the image load
// from / store to a dump area behind the tensor instead of being masked off.
// OUT6 (the head conv of a dtype f16mx6 ctx; ReLU layer): a.out takes the mx6 format of conv_trunk6.hip.  Not pipelined: swap of
// the lane halves, ReLU, conversion and the eight stores follow each tile's last plane step (one launch in 21).
template <bool RELU, bool RES, int NOMEM, bool OUT6 = false>
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

    const unsigned plane_bytes = (unsigned)(H * W * 64);   // the launcher keeps the whole tensor below 4 GiB: 32-bit offsets

    // ---- resident weights: 72 DMA instructions, 9 per wave
    {
        const unsigned char* wsrc = a.wpack + (size_t)half * TK::W_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(wsrc + k * 1024), (SR_AS3 void*)(wl + k * 1024), 16, 0, 0);
        }
    }

    // ---- per-lane plan of the plane-tile DMA.  42 instructions of 1 KiB: wave w issues k = w + 8 i, i < 5, and a sixth
    // one: k = w + 40 for waves 0, 1; the other waves repeat their fifth (same bytes to the same place) so that no wave
    // branches
    int kq[TK::PER_WAVE];
    int goff[TK::PER_WAVE];        // byte offset of the unit's source from the halo origin (row y0-1, column x0-1)
    int pyx[TK::PER_WAVE];         // halo column | halo row << 8; row 2^20: no such unit
#pragma unroll
    for (int i = 0; i < TK::PER_WAVE; ++i) {
        kq[i] = wave + 8 * i < TK::NINST ? wave + 8 * i : wave + 8 * (i - 1);
        const int u = kq[i] * 64 + lane;
        const int p = u >> 2, sl = (u & 3) ^ ((p >> 2) & 3);     // source unit of LDS unit (u & 3) of pixel p
        const int py = p / TK::HC, px = p - py * TK::HC;
        const bool exists = u < TK::NUNIT;
        goff[i] = (py * W + px) * 64 + sl * 16;
        pyx[i] = exists ? (px | (py << 8)) : (1 << 28);
        asm volatile("" : "+v"(goff[i]), "+v"(pyx[i]));          // keep them in registers: recomputing costs a division per DMA
    }
    const unsigned zoff = a.zero_off + (lane & 3) * 16;          // 64 zero bytes behind the tensor: halo outside the image
    const unsigned dump = a.zero_off + 256 + lane * 32;          // where lanes outside the image load / store
    struct Tile {
        unsigned off0;      // offset of the halo origin of plane 0 (wraps for row -1 / column -1: only in-image units use it)
        unsigned okm;       // bit i: this lane's unit of DMA instruction i lies inside the image
        unsigned eoff[2];   // epilogue: offset of this lane's record of (M block, first plane of this half), or `dump`
        unsigned pix6;      // OUT6: offset of pixel (M block h, column r) in half plane 0 of plane 0 of the mx6 output, or ~0u
    };
    // + one plane for the second plane of the half; a lane whose pixel lies outside the image stays in the dump area (M
    // block 1 of a tile that straddles the right edge is outside for lanes whose M block 0 pixel is inside)
    auto estep = [&](unsigned eoff) { return eoff != dump ? plane_bytes : 0u; };
    auto locate = [&](int tt) {
        Tile q;
        const int b = tt / per_img;
        const int rem = tt - b * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int y0 = ty * TK::TH - 1, x0 = tx * TK::TWX - 1;
        q.off0 = (unsigned)(b * 4) * plane_bytes + (unsigned)((y0 * W + x0) * 64);
        q.okm = 0;
#pragma unroll
        for (int i = 0; i < TK::PER_WAVE; ++i) {
            const unsigned gy = (unsigned)(y0 + (pyx[i] >> 8)), gx = (unsigned)(x0 + (pyx[i] & 255));
            q.okm |= ((gy < (unsigned)H) & (gx < (unsigned)W)) ? (1u << i) : 0u;   // a unit that does not exist fails the row test
        }
        const int y = y0 + 1 + wave;
        const unsigned row = (unsigned)(b * 4 + 2 * half) * plane_bytes + (unsigned)(y * W) * 64u + 32u * h;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int x = x0 + 1 + mb * 32 + r;
            const bool ok = y < H && x < W;
            q.eoff[mb] = ok ? row + (unsigned)x * 64u : dump;
        }
        if constexpr (OUT6) {
            const int x = x0 + 1 + 32 * h + r;
            q.pix6 = (y < H && x < W) ? (unsigned)(b * 4) * plane_bytes + (unsigned)(y * W + x) * 32u : ~0u;
        } else {
            q.pix6 = 0u;
        }
        return q;
    };
    // addresses are a.in + 32-bit offset: SGPR base + one VGPR per DMA
    auto issue = [&](const Tile& q, int c, int buf) {
        if (NOMEM & 1) return;
        const unsigned offc = q.off0 + (unsigned)c * plane_bytes;
        unsigned char* dst = abuf + buf * TK::A_BYTES;
#pragma unroll
        for (int i = 0; i < TK::PER_WAVE; ++i) {
            const unsigned off = ((q.okm >> i) & 1u) ? offc + (unsigned)goff[i] : zoff;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.in + off), (SR_AS3 void*)(dst + kq[i] * 1024), 16, 0, 0);
        }
    };
    Tile cur = locate(t);
    issue(cur, 0, 0);

    // the bias of this lane's 16 channels: lane (pixel r, h) owns channels 32 half + 8 g + 4 h + (0..3)
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + 32 * half + 8 * g + 4 * h);

    // B operand = pixels: lane (r, h) reads unit 2h (hi) and 2h+1 (lo) of halo pixel (wave + ky, 32 mb + r + kx)
    auto a_addr = [&](int ky, int kx, int mb) {
        const int p = (wave + ky) * TK::HC + mb * 32 + r + kx;
        return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4);
    };
    const unsigned char* const bl = wl + lane * 16;

    // One piece of a finished tile's epilogue: M block mb, plane pl of this half = 8 channels of this lane = 4 packed pairs.
    // ep_load issues the residual loads (inline asm: the compiler would wait vmcnt(0) for an ordinary load while DMA is in
    // flight, draining the prefetch), ep_pair computes pair k, ep_store stores; the plane step spreads them over its taps.
    struct Piece {
        u32x4 rh, rl, oh, ol;
        unsigned off;
    };
    auto ep_load = [&](Piece& pc, unsigned off) {
        pc.off = off;
        if (RES && !(NOMEM & 4)) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pc.rh) : "v"(off), "s"(a.res) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(pc.rl) : "v"(off), "s"(a.res) : "memory");
        } else {
            pc.rh = pc.rl = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // The residual loads of a piece are issued one plane step ahead (a load takes about 1.3 us under this kernel's traffic, a
    // third of a step is 0.65 us: waited for in the step that issued them, every step of a residual layer stalled on them --
    // tools/probe/trunk_ablate_b.py).  By the top of the next step they have landed: they are older than the DMA that step waits for.
    auto ep_landed = [&](Piece& pc) {
        if (RES && !(NOMEM & 4)) asm volatile("" : "+v"(pc.rh), "+v"(pc.rl));
    };
    auto ep_pair = [&](Piece& pc, float v0, float v1, int k) {
        if (RELU) {      // one v_max each (fmaxf() puts a canonicalising v_max in front: the accumulators may hold any bits); the
                         // builtin, not inline asm: hipcc keeps VALU instructions clear of in-flight MFMAs only for what it can see
            v0 = __builtin_amdgcn_fmed3f(v0, 0.f, __builtin_inff());
            v1 = __builtin_amdgcn_fmed3f(v1, 0.f, __builtin_inff());
        }
        if (RES) {
            v0 += bf_even(pc.rh[k]) + bf_even(pc.rl[k]);
            v1 += bf_odd(pc.rh[k]) + bf_odd(pc.rl[k]);
        }
        pc.oh[k] = pack_bf16(v0, v1);
        pc.ol[k] = pack_bf16(v0 - bf_even(pc.oh[k]), v1 - bf_odd(pc.oh[k]));
    };
    auto ep_store = [&](Piece& pc) {
        if (!(NOMEM & 4)) {
            *reinterpret_cast<u32x4*>(a.out + pc.off) = pc.oh;
            *reinterpret_cast<u32x4*>(a.out + pc.off + 16) = pc.ol;
        } else {
            asm volatile("" ::"v"(pc.oh), "v"(pc.ol));
        }
    };

    f32x16 acc[2], fin[2];      // running tile; finished tile whose epilogue is being drained
    Piece pcs[2];               // epilogue piece of this plane step / of the next one (its residual in flight)
    ep_load(pcs[0], dump);      // piece 0 of the (empty) tile that drains under the first tile
    f32x4 shp[2][2] = {};       // NOMEM & 8 only
    Tile done;                  // where `fin` goes
    done.eoff[0] = done.eoff[1] = dump;
#pragma unroll
    for (int i = 0; i < 16; ++i) fin[0][i] = fin[1][i] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weights and the first plane tile
    int q = 0;                  // running count of (tile, plane) steps: buffer = q & 1
    for (;;) {
        const int tn = t + npairs;
        const bool have_next = tn < t_end;
        const Tile nxt = locate(have_next ? tn : t);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][4 * g + j] = acc[1][4 * g + j] = bias4[g][j];

        auto plane_step = [&](auto cc) {
            constexpr int C = decltype(cc)::value;
            constexpr int mbp = C >> 1, plp = C & 1;    // the piece of the finished tile that drains under this step
            // The DMA of this step has landed: vmcnt counts in issue order and the only younger operations are the two
            // stores of the previous step's epilogue piece (none in the ablation build without them).
            if ((NOMEM & 4) || (OUT6 && C != 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (OUT6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the eight stores behind the previous tile
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // every wave's part landed; every wave is done reading the other buffer
            asm volatile("" ::: "memory");
            Piece& pc = pcs[C & 1];
            if constexpr (!OUT6) {   // the next step's piece: C + 1 of the finished tile, or piece 0 of the tile being computed (finished by then)
                constexpr int CN = (C + 1) & 3;
                const unsigned base = C < 3 ? done.eoff[CN >> 1] : cur.eoff[0];
                ep_load(pcs[(C + 1) & 1], base + ((CN & 1) ? estep(base) : 0u));   // before the DMA: older in the vmcnt queue
            }
            if (C < 3) issue(cur, C + 1, (q + 1) & 1);
            else issue(nxt, 0, (q + 1) & 1);        // past the last tile: re-reads this tile's first plane, harmless
            const unsigned char* at = abuf + (q & 1) * TK::A_BYTES;
            const unsigned char* bw = bl + C * (9 * 2 * 1024);
            ++q;
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
            if (!(NOMEM & 2)) load_tap(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = tap & 1;
                if (!(NOMEM & 2)) {
                    if (tap + 1 < 9) load_tap(tap + 1, s ^ 1);
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, fb[s][0]), wlo = __builtin_bit_cast(bf16x8, fb[s][1]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const bf16x8 xh = __builtin_bit_cast(bf16x8, fa[s][mb][0]), xl = __builtin_bit_cast(bf16x8, fa[s][mb][1]);
                        if constexpr (NOMEM & 8) {
                            // MFMA-shape probe of the micro-benchmark (WRONG results): the same operands, registers, LDS reads and
                            // MFMA cycles as two v_mfma_f32_16x16x32_bf16 per v_mfma_f32_32x32x16_bf16 -- the clock the chip
                            // holds depends on the shape (MI355X_MICROARCH.md, DVFS give-back item 7)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                shp[mb][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, shp[mb][e], 0, 0, 0);
                                shp[mb][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, shp[mb][e], 0, 0, 0);
                                shp[mb][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, shp[mb][e], 0, 0, 0);
                            }
                        } else {
                            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[mb], 0, 0, 0);
                            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh, acc[mb], 0, 0, 0);
                            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[mb], 0, 0, 0);
                        }
                    }
                }
                // the epilogue piece, spread over the taps: pair k under tap 3 + k, the stores under tap 7
                if constexpr (!OUT6) {
                    if (tap == 3) ep_landed(pc);
                    if (tap >= 3 && tap < 7) {
                        const int k = tap - 3;
                        ep_pair(pc, fin[mbp][8 * plp + 2 * k], fin[mbp][8 * plp + 2 * k + 1], k);
                    }
                    if (tap == 7) ep_store(pc);
                }
                // one fragment read of the next tap and a few VALU / DMA instructions behind each MFMA
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (tap == 0) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    if (tap == 0) __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        plane_step(std::integral_constant<int, 0>{});
        plane_step(std::integral_constant<int, 1>{});
        plane_step(std::integral_constant<int, 2>{});
        plane_step(std::integral_constant<int, 3>{});
        if constexpr (NOMEM & 8) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[0][i] += shp[0][i & 1][i >> 2];
                acc[1][i] += shp[1][i & 1][i >> 2];
            }
        }
        if constexpr (OUT6) {      // this tile's epilogue, here and now, in the mx6 format (conv_trunk6.hip)
            mx6::f32x16 V0, V1;
            mx6::swap_halves(acc[0], acc[1], V0, V1);
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    V0[i] = __builtin_amdgcn_fmed3f(V0[i], 0.f, __builtin_inff());
                    V1[i] = __builtin_amdgcn_fmed3f(V1[i], 0.f, __builtin_inff());
                }
            }
            mx6::Rec6 rec;
            mx6::encode6(V0, V1, rec);
            const unsigned hplane = plane_bytes >> 1;      // the output has the input's geometry: half planes of H x W x 32 bytes
            auto uoff = [&](int pl, int u) { return cur.pix6 != ~0u ? cur.pix6 + (unsigned)(2 * pl + (u >> 1)) * hplane + 16u * (u & 1) : dump + 16u * (u & 1); };
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(a.out + uoff(half, k)) = rec.hi[k];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                *reinterpret_cast<u32x4*>(a.out + uoff(2, 2 * half + k)) = rec.x6[k];
                *reinterpret_cast<u32x4*>(a.out + uoff(3, 2 * half + k)) = rec.lo6[k];
            }
        }
        fin[0] = acc[0];
        fin[1] = acc[1];
        done = cur;
        if (!have_next) break;
        t = tn;
        cur = nxt;
    }
    if constexpr (OUT6) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // the last tile's epilogue has no MFMAs to hide under
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            Piece& pc = pcs[0];
            if (mb + pl > 0) ep_load(pc, done.eoff[mb] + (pl ? estep(done.eoff[mb]) : 0u));   // piece 0 came with the last plane step
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ep_landed(pc);
#pragma unroll
            for (int k = 0; k < 4; ++k) ep_pair(pc, fin[mb][8 * pl + 2 * k], fin[mb][8 * pl + 2 * k + 1], k);
            ep_store(pc);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing DMA (into LDS) must not outlive the workgroup
}

// ---------------------------------------------------------------------------------------------------------------
// conv2d_transpose_1 (reference layers.py:73: Conv2DTranspose 64 -> 64, 3x3, strides 2, 'same', ReLU) in the same style: split-planar
// in (the trunk's output) and out (what conv2d_transpose_2's staging copies into LDS as it is).
//   out[2i + ky, 2j + kx] += in[i, j] * w[ky, kx]  (full result minus its last row / column)
// i.e. output parity (py, px) of input pixel (i, j) collects the taps with ky = py (mod 2), kx = px (mod 2) from the pixels
// (i - ky/2, j - kx/2): 4 / 2 / 2 / 1 taps, nine per input pixel, no zero insertion.  A wave owns one row of 32 input pixels of an
// 8 x 32 tile and keeps the four parity accumulators (32 channels of this workgroup's half) alive while K runs plane by plane;
// per plane four shifted pixel fragments (hi, lo) feed 27 MFMAs.  Weights of the half (72 KiB) resident, 9 x 33 plane tiles
// (19 KiB) double buffered by DMA; the epilogue of a finished tile (4 parities x 2 planes = 8 pieces: ReLU, hi / lo split, two
// 16-byte stores) drains two pieces per plane step inside the next tile's MFMA stream.
struct TKT {
    static constexpr int TH = 8, TWX = 32;
    static constexpr int HR = TH + 1, HC = TWX + 1;          // halo: one row above, one column to the left
    static constexpr int NPIX = HR * HC;                     // 297
    static constexpr int NUNIT = NPIX * 4;
    static constexpr int NINST = (NUNIT + 63) / 64;          // 19 DMA wave-instructions
    static constexpr int PER_WAVE = (NINST + 7) / 8;         // 3 (waves 0..2) or 2
    static constexpr int A_BYTES = NINST * 1024;             // 19,456 per buffer
    static constexpr int SCR_OFF = TK::W_BYTES + 2 * A_BYTES;   // per-wave 4 KiB store transpose scratch
    static constexpr int LDS = SCR_OFF + 8 * 4096;           // 145,408
};

template <int NOMEM>
__global__ __launch_bounds__(512, 2) void trunk_convT_kernel(const TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wl = smem;
    unsigned char* const abuf = smem + TK::W_BYTES;

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
    const unsigned oplane = 4u * plane_bytes;                 // output plane: 2H x 2W records
    const unsigned orow = (unsigned)(2 * W) * 64u;

    {   // resident weights: [plane][tap][hi, lo][lane] of this half
        const unsigned char* wsrc = a.wpack + (size_t)half * TK::W_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = wave + 8 * i;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(wsrc + k * 1024), (SR_AS3 void*)(wl + k * 1024), 16, 0, 0);
        }
    }
    int goff[TKT::PER_WAVE];
    int pyx[TKT::PER_WAVE];
#pragma unroll
    for (int i = 0; i < TKT::PER_WAVE; ++i) {
        const int k = wave + 8 * i < TKT::NINST ? wave + 8 * i : wave + 8 * (i - 1);
        const int u = k * 64 + lane;
        const int p = u >> 2, sl = (u & 3) ^ ((p >> 2) & 3);
        const int py = p / TKT::HC, px = p - py * TKT::HC;
        goff[i] = (py * W + px) * 64 + sl * 16;
        pyx[i] = u < TKT::NUNIT ? (px | (py << 8)) : (1 << 28);
        asm volatile("" : "+v"(goff[i]), "+v"(pyx[i]));
    }
    const unsigned zoff = a.zero_off + (lane & 3) * 16;
    const unsigned dump = a.out_tail + 256 + lane * 16;      // store lanes outside the image write behind the OUTPUT tensor
    struct Tile {
        unsigned off0, okm;
        unsigned erow;   // output: offset of record (row 2 iy, column 2 x0) of the half's first plane + this lane's 16 bytes
        int xlim;        // valid input columns from x0 on (0: the wave's row lies outside the image)
    };
    auto locate = [&](int tt) {
        Tile q;
        const int b = tt / per_img;
        const int rem = tt - b * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int y0 = ty * TKT::TH - 1, x0 = tx * TKT::TWX - 1;
        q.off0 = (unsigned)(b * 4) * plane_bytes + (unsigned)((y0 * W + x0) * 64);
        q.okm = 0;
#pragma unroll
        for (int i = 0; i < TKT::PER_WAVE; ++i) {
            const unsigned gy = (unsigned)(y0 + (pyx[i] >> 8)), gx = (unsigned)(x0 + (pyx[i] & 255));
            q.okm |= ((gy < (unsigned)H) & (gx < (unsigned)W)) ? (1u << i) : 0u;
        }
        const int iy = y0 + 1 + wave;
        q.erow = (unsigned)(b * 4 + 2 * half) * oplane + (unsigned)(2 * iy) * orow + (unsigned)(2 * (x0 + 1)) * 64u + (unsigned)lane * 16u;
        q.xlim = iy < H ? W - (x0 + 1) : 0;
        return q;
    };
    auto issue = [&](const Tile& q, int c, int buf) {
        if (NOMEM & 1) return;
        const unsigned offc = q.off0 + (unsigned)c * plane_bytes;
        unsigned char* dst = abuf + buf * TKT::A_BYTES;
#pragma unroll
        for (int i = 0; i < TKT::PER_WAVE; ++i) {
            const int k = wave + 8 * i < TKT::NINST ? wave + 8 * i : wave + 8 * (i - 1);
            const unsigned off = ((q.okm >> i) & 1u) ? offc + (unsigned)goff[i] : zoff;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.in + off), (SR_AS3 void*)(dst + k * 1024), 16, 0, 0);
        }
    };
    Tile cur = locate(t);
    issue(cur, 0, 0);

    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + 32 * half + 8 * g + 4 * h);

    // pixel fragment of shift (dy, dx): halo pixel (wave + 1 - dy, r + 1 - dx), hi unit 2h, lo unit 2h + 1 (swizzled)
    auto x_addr = [&](int dy, int dx) {
        const int p = (wave + 1 - dy) * TKT::HC + r + 1 - dx;
        return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4);
    };
    const unsigned char* const bl = wl + lane * 16;

    // Store transpose.  Lane (r, h) holds, for one output row parity and plane, the hi / lo units of output pixels 2r and 2r + 1:
    // stored from there, one instruction puts 32 bytes into each of 32 different 128-byte lines and a line needs four
    // instructions -- the write side then takes 0.36 ms alone (micro-benchmark without MFMAs) against 0.19 ms for everything
    // else.  So the wave passes its 4 KiB (64 records = one contiguous stretch of the output row) through a private LDS
    // scratch -- logical 16-byte piece P = 8 r + 4 px + 2 h + hl, stored at block P >> 3 with the unit index XORed by the block
    // number (writes and reads both conflict free) -- and stores piece 64 k + lane in instruction k: 1 KiB contiguous each.
    unsigned char* const scr = smem + TKT::SCR_OFF + wave * 4096;
    unsigned sw[4], sr4[4];     // scratch addresses: this lane's four pieces to write, the four pieces it stores
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int unit_w = 4 * (k >> 1) + 2 * h + (k & 1);                 // k = 2 px + hl
        sw[k] = (unsigned)(r * 128 + ((unit_w ^ (r & 7)) << 4));
        const int P = 64 * k + lane, blk = P >> 3;
        sr4[k] = (unsigned)(blk * 128 + (((P & 7) ^ (blk & 7)) << 4));
    }
    asm volatile("" : "+v"(sw[0]), "+v"(sw[1]), "+v"(sw[2]), "+v"(sw[3]), "+v"(sr4[0]), "+v"(sr4[1]), "+v"(sr4[2]), "+v"(sr4[3]));

    auto ep_pair = [&](float v0, float v1, u32x4& oh, u32x4& ol, int k) {
        v0 = __builtin_amdgcn_fmed3f(v0, 0.f, __builtin_inff());      // ReLU, one instruction (see trunk_conv_kernel)
        v1 = __builtin_amdgcn_fmed3f(v1, 0.f, __builtin_inff());
        const unsigned hi = pack_bf16(v0, v1);
        oh[k] = hi;
        ol[k] = pack_bf16(v0 - bf_even(hi), v1 - bf_odd(hi));
    };
    auto scr_write = [&](int px, const u32x4& oh, const u32x4& ol) {
        *reinterpret_cast<u32x4*>(scr + sw[2 * px + 0]) = oh;
        *reinterpret_cast<u32x4*>(scr + sw[2 * px + 1]) = ol;
    };
    // the wave's 64 records of output row parity py, plane pl: four contiguous 1 KiB stores
    auto scr_store = [&](unsigned erow, int xlim, int py, int pl) {
        const unsigned base = erow + (unsigned)py * orow + (unsigned)pl * oplane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(scr + sr4[k]);
            const unsigned off = (8 * k + (lane >> 3)) < xlim ? base + (unsigned)k * 1024u : dump;
            if (!(NOMEM & 4)) *reinterpret_cast<u32x4*>(a.out + off) = v;
            else asm volatile("" ::"v"(v));
        }
    };

    f32x16 acc[4], fin[4];      // by output parity 2 py + px
    unsigned done_erow = 0;
    int done_xlim = 0;
#pragma unroll
    for (int pq = 0; pq < 4; ++pq)
#pragma unroll
        for (int i = 0; i < 16; ++i) fin[pq][i] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int q = 0;
    for (;;) {
        const int tn = t + npairs;
        const bool have_next = tn < t_end;
        Tile nxt;
#pragma unroll
        for (int pq = 0; pq < 4; ++pq)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[pq][4 * g + j] = bias4[g][j];

        auto plane_step = [&](auto cc) {
            constexpr int C = decltype(cc)::value;
            // what drains of the finished tile under this step: output row parity C >> 1, plane C & 1 of the half, both column parities
            constexpr int PY = C >> 1, PLN = C & 1;
            if (NOMEM & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // younger: the four stores of the previous step
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (C < 3) {
                issue(cur, C + 1, (q + 1) & 1);
            } else {
                nxt = locate(have_next ? tn : t);
                issue(nxt, 0, (q + 1) & 1);
            }
            const unsigned char* at = abuf + (q & 1) * TKT::A_BYTES;
            const unsigned char* bw = bl + C * (9 * 2 * 1024);
            ++q;
            u32x4 oh0, ol0, oh1, ol1;
            // tap order: the four taps of shift (0, 0) first (its fragment and tap 0's weights are the only exposed reads), then
            // shift (0, 1), (1, 0), (1, 1); tap (ky, kx) = shift (ky >> 1, kx >> 1), parity 2 (ky & 1) + (kx & 1)
            constexpr int order[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8};
            u32x4 xs[4][2], wf[2][2];
            auto load_x = [&](int sft) {
                const int ad = x_addr(sft >> 1, sft & 1);
                xs[sft][0] = *reinterpret_cast<const u32x4*>(at + ad);
                xs[sft][1] = *reinterpret_cast<const u32x4*>(at + (ad ^ 16));
            };
            auto load_w = [&](int tap, int sb) {
                wf[sb][0] = *reinterpret_cast<const u32x4*>(bw + (tap * 2 + 0) * 1024);
                wf[sb][1] = *reinterpret_cast<const u32x4*>(bw + (tap * 2 + 1) * 1024);
            };
            if (!(NOMEM & 2)) {
                load_x(0);
                load_w(order[0], 0);
            }
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                const int tap = order[n], ky = tap / 3, kx = tap - 3 * ky;
                const int sft = (ky >> 1) * 2 + (kx >> 1), par = (ky & 1) * 2 + (kx & 1), sb = n & 1;
                if (!(NOMEM & 2)) {
                    if (n + 1 < 9) load_w(order[n + 1], sb ^ 1);
                    if (n == 0) load_x(1);
                    if (n == 1) load_x(2);
                    if (n == 2) load_x(3);
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[sb][0]), wlo = __builtin_bit_cast(bf16x8, wf[sb][1]);
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, xs[sft][0]), xl = __builtin_bit_cast(bf16x8, xs[sft][1]);
                    acc[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[par], 0, 0, 0);
                    acc[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh, acc[par], 0, 0, 0);
                    acc[par] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[par], 0, 0, 0);
                }
                // epilogue: column parity 0 under taps 0..3 (into the scratch under tap 4), parity 1 under taps 4..7, read back
                // and stored under tap 8
                if (n < 4) ep_pair(fin[2 * PY][8 * PLN + 2 * n], fin[2 * PY][8 * PLN + 2 * n + 1], oh0, ol0, n);
                if (n == 4) scr_write(0, oh0, ol0);
                if (n >= 4 && n < 8) ep_pair(fin[2 * PY + 1][8 * PLN + 2 * (n - 4)], fin[2 * PY + 1][8 * PLN + 2 * (n - 4) + 1], oh1, ol1, n - 4);
                if (n == 8) {
                    scr_write(1, oh1, ol1);
                    scr_store(done_erow, done_xlim, PY, PLN);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (n + 1 < 9 && k < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (n < 3 && k == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (n == 0) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    if (n == 4 && k < 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (n == 8 && k == 0) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                    if (n == 8 && k == 1) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    if (n == 0) __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                    if (n == 8 && k == 2) __builtin_amdgcn_sched_group_barrier(0x010, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        plane_step(std::integral_constant<int, 0>{});
        plane_step(std::integral_constant<int, 1>{});
        plane_step(std::integral_constant<int, 2>{});
        plane_step(std::integral_constant<int, 3>{});
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) fin[pq] = acc[pq];
        done_erow = cur.erow;
        done_xlim = cur.xlim;
        if (!have_next) break;
        t = tn;
        cur = nxt;
    }
    // the last tile's epilogue has no MFMAs to hide under
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                u32x4 oh, ol;
#pragma unroll
                for (int k = 0; k < 4; ++k) ep_pair(fin[2 * py + px][8 * pl + 2 * k], fin[2 * py + px][8 * pl + 2 * k + 1], oh, ol, k);
                scr_write(px, oh, ol);
            }
            scr_store(done_erow, done_xlim, py, pl);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

template <bool RELU, bool RES, int NOMEM, bool OUT6 = false>
static hipError_t launch_trunk_t(const TrunkArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trunk_conv_kernel<RELU, RES, NOMEM, OUT6>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TK::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((trunk_conv_kernel<RELU, RES, NOMEM, OUT6>), dim3(grid), dim3(512), TK::LDS, s, a);
    return hipGetLastError();
}

template <int NOMEM>
static hipError_t launch_trunkT_t(const TrunkArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trunk_convT_kernel<NOMEM>), hipFuncAttributeMaxDynamicSharedMemorySize, TKT::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((trunk_convT_kernel<NOMEM>), dim3(grid), dim3(512), TKT::LDS, s, a);
    return hipGetLastError();
}

// conv2d_transpose_1 on split-planar tensors: a.H, a.W = INPUT size, output 2H x 2W; a.zero_off / a.out_tail = byte offsets of the
// zero / dump tails behind the input / output tensors
hipError_t launch_trunk_convT(TrunkArgs a, hipStream_t s) {
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.out || !a.wpack || !a.bias || a.res || !a.relu) return hipErrorInvalidValue;
    if ((double)a.B * a.H * a.W * 1024.0 + 4096.0 > 4294967295.0 || (double)a.zero_off + 4096.0 > 4294967295.0 ||
        (double)a.out_tail + 4096.0 > 4294967295.0)
        return hipErrorInvalidValue;
    a.tiles_x = (a.W + TKT::TWX - 1) / TKT::TWX;
    a.tiles_y = (a.H + TKT::TH - 1) / TKT::TH;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    int npairs = std::min(std::max(1, trunk_num_cus() / 16), per_xcd);
    const int rounds = (per_xcd + npairs - 1) / npairs;
    npairs = (per_xcd + rounds - 1) / rounds;
    const int grid = 16 * npairs;
    switch (a.dbg & 7) {
        case 0: return launch_trunkT_t<0>(a, grid, s);
        case 1: return launch_trunkT_t<1>(a, grid, s);
        case 2: return launch_trunkT_t<2>(a, grid, s);
        case 4: return launch_trunkT_t<4>(a, grid, s);
        case 5: return launch_trunkT_t<5>(a, grid, s);
        default: return launch_trunkT_t<7>(a, grid, s);
    }
}

hipError_t launch_trunk_conv(TrunkArgs a, hipStream_t s) {
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.out || !a.wpack || !a.bias) return hipErrorInvalidValue;
    // 32-bit offsets: the tensor, its 256 zero bytes and the 2 KiB dump area behind them
    if ((double)a.B * a.H * a.W * 256.0 + 4096.0 > 4294967295.0 || (double)a.zero_off + 4096.0 > 4294967295.0) return hipErrorInvalidValue;
    if ((a.relu != 0) == (a.res != nullptr)) return hipErrorInvalidValue;   // the trunk has ReLU layers and residual layers, nothing else
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
    const int grid = 16 * npairs;
    if (a.dbg) {   // micro-benchmark ablations (tools/trunk_ablate.sh), residual layer only
        if ((a.dbg & 15) == 8) return launch_trunk_t<false, true, 8>(a, grid, s);     // MFMA-shape probe, whole kernel
        if ((a.dbg & 15) == 13) return launch_trunk_t<false, true, 13>(a, grid, s);   // MFMA-shape probe, MFMAs + LDS reads only
        switch (a.dbg & 7) {
            case 1: return launch_trunk_t<false, true, 1>(a, grid, s);
            case 2: return launch_trunk_t<false, true, 2>(a, grid, s);
            case 3: return launch_trunk_t<false, true, 3>(a, grid, s);
            case 4: return launch_trunk_t<false, true, 4>(a, grid, s);
            case 5: return launch_trunk_t<false, true, 5>(a, grid, s);
            case 6: return launch_trunk_t<false, true, 6>(a, grid, s);
            default: return launch_trunk_t<false, true, 7>(a, grid, s);
        }
    }
    if (a.out_mx6) return a.res ? hipErrorInvalidValue : launch_trunk_t<true, false, 0, true>(a, grid, s);   // the head conv of an f16mx6 ctx
    return a.res ? launch_trunk_t<false, true, 0>(a, grid, s) : launch_trunk_t<true, false, 0>(a, grid, s);
}

}  // namespace sr
