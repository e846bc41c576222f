// SRNet's x4 tail in the style of conv_trunk.hip: conv2d_transpose_2 (64 -> 64, stride 2, ReLU) + conv2d_36 (64 -> 3) + clip
// (reference layers.py:74-75) on the split-planar output of conv2d_transpose_1, K running plane by plane.
//
// convT3x3_gather_kernel (conv_mfma.hip) walks K tap by tap: it needs the whole 64-channel input tile at once (47.5 KiB through
// registers into LDS per 128 pixels), re-fetches the 147 KiB of weights per 128 pixels and synchronises nine times per tile; its
// input staging and its stores add to the MFMA time instead of hiding under it (micro-benchmark ablations: 0.90 ms of 1.22 without
// them).  Here, as in trunk_convT_kernel:
//   * one persistent 512-thread workgroup per CU, tile = 8 rows x 32 columns of input pixels (16 x 64 output pixels), wave w = row w;
//   * a wave keeps ALL its accumulators alive -- 4 output parities x 64 channels = 128 registers -- while K runs over the four
//     16-channel planes; per plane the 9 x 33 halo tile (19 KiB) and that plane's weights for all 64 outputs (36 KiB: both halves of
//     the trunkT packing) arrive by global -> LDS DMA, double buffered, one barrier per plane: 144 KiB of weights per 256 pixels,
//     no registers, no VALU, no ds_write on the way;
//   * then conv2d_36's 64 x 27 tap projection on the accumulators (12 MFMAs per parity, operands from LDS), and its nine-tap sum in
//     LDS exactly as convT3x3_gather_kernel does it -- interior pixels finished and stored, the tile's outermost ring and the ring
//     around it as per-tile partial sums for gather_fix_kernel -- in two passes of nine record rows, because the next tile's first
//     plane is already landing in the other buffer pair.
// LDS map (160 KiB): [W0 36 K][A0 19 K][F 50 K][A1 19 K][W1 36 K]; plane c uses pair c & 1, so the records of a finished tile take
// F (behind the 8 KiB of projection weights) + A1 + W1 while pair 0 receives the next tile.
// Sums run plane-major instead of tap-major: results differ from convT3x3_gather_kernel by the association of fp32 sums (1e-6).
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace sr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SR_AS1 __attribute__((address_space(1)))
#define SR_AS3 __attribute__((address_space(3)))

namespace {

struct TL {
    static constexpr int TH = 8, TWX = 32;                   // input tile; output tile 16 x 64
    static constexpr int HR = TH + 1, HC = TWX + 1;          // halo: one row above, one column to the left
    static constexpr int NPIX = HR * HC;                     // 297
    static constexpr int NUNIT = NPIX * 4;
    static constexpr int A_INST = (NUNIT + 63) / 64;         // 19 DMA wave-instructions per plane tile
    static constexpr int A_PER_WAVE = (A_INST + 7) / 8;      // 3
    static constexpr int A_BYTES = A_INST * 1024;            // 19,456
    static constexpr int W_INST = 2 * 9 * 2;                 // 36: [half][tap][hi, lo] of one plane
    static constexpr int W_PER_WAVE = (W_INST + 7) / 8;      // 5
    static constexpr int W_BYTES = W_INST * 1024;            // 36,864
    static constexpr int W_HALF_STRIDE = 4 * 9 * 2 * 1024;   // trunkT packing: [half][plane][tap][hi, lo][lane]
    static constexpr int W_PLANE_STRIDE = 9 * 2 * 1024;
    static constexpr int OFF_W0 = 0, OFF_A0 = W_BYTES, OFF_F = OFF_A0 + A_BYTES;
    static constexpr int LDS = 160 * 1024;
    static constexpr int OFF_W1 = LDS - W_BYTES, OFF_A1 = OFF_W1 - A_BYTES;
    static constexpr int PROJ_BYTES = 8 * 1024;              // 64 x 32 projection matrix, hi and lo, as MFMA A fragments
    static constexpr int PTS = 29;                           // floats per gather record (27 used; odd: conflict-free column reads)
    static constexpr int REC_ROWS = 9;
    static constexpr int OFF_BIAS = OFF_F + PROJ_BYTES;      // 64 floats
    static constexpr int OFF_ZREC = OFF_BIAS + 256;          // one record of zeros: what a tap outside the tile reads
    static constexpr int OFF_REC = OFF_ZREC + 128;
    static constexpr int REC_BYTES = REC_ROWS * 64 * PTS * 4;
    static_assert(OFF_REC + REC_BYTES <= LDS, "records fit behind the projection weights");
    static_assert(OFF_F + PROJ_BYTES <= OFF_A1, "projection weights stay clear of the plane buffers");
};

__device__ __forceinline__ unsigned pack_bf16_2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

}  // namespace

template <int DBG>
__global__ __launch_bounds__(512, 2) void tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;

    // workgroup -> (XCD, slot): ids go round-robin over the 8 XCDs; every XCD owns one contiguous band of tiles
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nslots = (int)(gridDim.x >> 3);
    const int per_img = a.tiles_x * a.tiles_y;
    const int ntiles = per_img * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    const int t_end = min(ntiles, (xcd + 1) * per_xcd);
    int t = xcd * per_xcd + slot;
    if (t >= t_end) return;

    const unsigned plane_bytes = (unsigned)(H * W * 64);

    // the projection weights, once: 8 KiB = one 16-byte vector per thread
    __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.proj_w + (size_t)tid * 16),
                                     (SR_AS3 void*)(smem + TL::OFF_F + wave * 1024), 16, 0, 0);

    // ---- DMA.  Plane tile: 19 instructions of 1 KiB, wave w issues k = w + 8 i; weights of a plane: 36 instructions.  A wave whose
    // last index does not exist repeats its previous one (same bytes to the same place): no branches in the plane body.  Nothing of
    // the plan is kept in registers between plane steps (a few dozen VALU per step against 54 MFMAs): the accumulators need them.
    auto ka = [&](int i) { return wave + 8 * i < TL::A_INST ? wave + 8 * i : wave + 8 * (i - 1); };
    auto kw = [&](int i) { return wave + 8 * i < TL::W_INST ? wave + 8 * i : wave + 8 * (i - 1); };
    struct Tile {
        int b, y0, x0;          // image, first input row / column of the tile
    };
    auto locate = [&](int tt) {
        Tile q;
        q.b = tt / per_img;
        const int rem = tt - q.b * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        q.y0 = ty * TL::TH;
        q.x0 = tx * TL::TWX;
        return q;
    };
    auto issue = [&](const Tile& q, int c) {      // plane c of tile q into buffer pair c & 1
        if (DBG & 4) return;
        unsigned char* adst = smem + ((c & 1) ? TL::OFF_A1 : TL::OFF_A0);
        unsigned char* wdst = smem + ((c & 1) ? TL::OFF_W1 : TL::OFF_W0);
        const int hy = q.y0 - 1, hx = q.x0 - 1;
        const unsigned offc = (unsigned)(q.b * 4 + c) * plane_bytes;
#pragma unroll
        for (int i = 0; i < TL::A_PER_WAVE; ++i) {
            const int u = ka(i) * 64 + lane;
            const int p = u >> 2, sl = (u & 3) ^ ((p >> 2) & 3);     // source unit of LDS unit (u & 3) of halo pixel p (XOR swizzle)
            const int py = p / TL::HC, px = p - py * TL::HC;
            const unsigned gy = (unsigned)(hy + py), gx = (unsigned)(hx + px);
            const bool ok = u < TL::NUNIT && gy < (unsigned)H && gx < (unsigned)W;
            const unsigned off = ok ? offc + (gy * (unsigned)W + gx) * 64u + (unsigned)sl * 16u : a.zero_off + (unsigned)(lane & 3) * 16u;
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(a.in + off), (SR_AS3 void*)(adst + ka(i) * 1024), 16, 0, 0);
        }
        const unsigned char* wsrc = a.wpack + (size_t)c * TL::W_PLANE_STRIDE + lane * 16;
#pragma unroll
        for (int i = 0; i < TL::W_PER_WAVE; ++i) {
            const int hf = kw(i) / 18, rest = kw(i) - 18 * hf;      // wave-uniform
            __builtin_amdgcn_global_load_lds((const SR_AS1 void*)(wsrc + hf * TL::W_HALF_STRIDE + rest * 1024),
                                             (SR_AS3 void*)(wdst + kw(i) * 1024), 16, 0, 0);
        }
    };
    // conv2d_transpose_2's bias through LDS (a per-tile global load would wait for the previous tile's stores: vmcnt is in order),
    // conv2d_36's in registers
    float* const lbias = reinterpret_cast<float*>(smem + TL::OFF_BIAS);
    if (tid < 64) lbias[tid] = a.bias[tid];
    else if (tid < 96) reinterpret_cast<float*>(smem + TL::OFF_ZREC)[tid - 64] = 0.f;
    float gb[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) gb[co] = co < a.gather_cout ? a.gbias[co] : 0.f;
    Tile cur = locate(t);
    issue(cur, 0);

    // pixel fragment of shift (dy, dx): halo pixel (wave + 1 - dy, r + 1 - dx), hi unit 2h, lo unit 2h + 1 (swizzled)
    auto x_addr = [&](int dy, int dx) {
        const int p = (wave + 1 - dy) * TL::HC + r + 1 - dx;
        return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4);
    };
    float* const PT = reinterpret_cast<float*>(smem + TL::OFF_REC);
    const unsigned char* const pwl = smem + TL::OFF_F + lane * 16;

    for (;;) {
        const int tn = t + nslots;
        const bool have_next = tn < t_end;
        Tile nxt = cur;
        f32x16 acc[4][2];      // [output parity 2 py + px][half of the channels]; lane (pixel r, h) owns channels 32 n + 8 g + 4 h + j
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + 32 * n + 8 * g + 4 * h);
#pragma unroll
                for (int pq = 0; pq < 4; ++pq)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[pq][n][4 * g + j] = b4[j];
            }

        // conv2d_36's tap projection: p[4 g + j] = P[pixel r][column 8 g + 4 h + j] (column = tap * 3 + channel) per parity; K = the
        // 64 activations of the pixel in the accumulators' own order (the host permutes the packed matrix to match).  One step =
        // 16 channels (half n, group pair s): ReLU, hi / lo split (VALU), three MFMAs.
        f32x16 pk[4];
#pragma unroll
        for (int pq = 0; pq < 4; ++pq)
#pragma unroll
            for (int i = 0; i < 16; ++i) pk[pq][i] = 0.f;
        auto proj_step = [&](int pq, int kidx) {
            if (DBG & 2) {
                if (kidx == 0) pk[pq] = acc[pq][0] + acc[pq][1];
                return;
            }
            const int n = kidx >> 1, s2 = kidx & 1;
            u32x4 xh4, xl4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i0 = 8 * s2 + 2 * e;      // k index 2e, 2e + 1 of step (n, s) = accumulator registers 8s + 2e, + 1
                float v0 = acc[pq][n][i0], v1 = acc[pq][n][i0 + 1];
                // ReLU as ONE instruction: fmaxf() costs a canonicalising v_max in front of the v_max (accumulators may hold any
                // bits); the med3 builtin compiles to a single v_max.  (NOT inline asm: hipcc does not see an asm's registers when
                // it spaces VALU instructions from the MFMAs around them -- a build with this phase inside wave-conditional branches
                // computed wrong projections for some waves with `asm("v_max_f32 ...")` and right ones with the builtin.)
                v0 = __builtin_amdgcn_fmed3f(v0, 0.f, __builtin_inff());
                v1 = __builtin_amdgcn_fmed3f(v1, 0.f, __builtin_inff());
                const unsigned hi = pack_bf16_2(v0, v1);
                xh4[e] = hi;
                xl4[e] = pack_bf16_2(v0 - __builtin_bit_cast(float, hi << 16), v1 - __builtin_bit_cast(float, hi & 0xFFFF0000u));
            }
            const bf16x8 wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pwl + (kidx * 2 + 0) * 1024));
            const bf16x8 wl2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pwl + (kidx * 2 + 1) * 1024));
            const bf16x8 xh = __builtin_bit_cast(bf16x8, xh4), xl = __builtin_bit_cast(bf16x8, xl4);
            pk[pq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, pk[pq], 0, 0, 0);
            pk[pq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl2, xh, pk[pq], 0, 0, 0);
            pk[pq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, pk[pq], 0, 0, 0);
        };

        auto plane_step = [&](auto cc) {
            constexpr int C = decltype(cc)::value;
            // this plane's tile and weights have landed (everything this wave issued is done; the barrier covers the other waves),
            // and every wave is done reading the other buffer pair
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (C < 3) {
                issue(cur, C + 1);
            } else {
                nxt = locate(have_next ? tn : t);      // past the last tile: re-reads this tile's first plane, harmless (no branch)
                issue(nxt, 0);
            }
            const unsigned char* at = smem + ((C & 1) ? TL::OFF_A1 : TL::OFF_A0);
            const unsigned char* bw = smem + ((C & 1) ? TL::OFF_W1 : TL::OFF_W0) + lane * 16;
            // tap (ky, kx) = shift (ky >> 1, kx >> 1), parity 2 (ky & 1) + (kx & 1); the four taps of shift (0, 0) first, then shift
            // (0, 1), (1, 0), (1, 1): two shifts' pixel fragments live at a time, shift s in buffer s & 1, read two taps before its first use
            constexpr int order[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8};
            // half-tap pipeline: step m = (tap order[m >> 1], channel half m & 1) = 3 MFMAs; its two weight fragments are read one step ahead
            u32x4 xs[2][2], wf[2][2];      // [shift & 1][hi, lo]; [m & 1][hi, lo]
            auto load_x = [&](int sft) {
                const int ad = x_addr(sft >> 1, sft & 1);
                xs[sft & 1][0] = *reinterpret_cast<const u32x4*>(at + ad);
                xs[sft & 1][1] = *reinterpret_cast<const u32x4*>(at + (ad ^ 16));
            };
            auto load_w = [&](int m) {
                const int tap = order[m >> 1], nb = m & 1;
                wf[m & 1][0] = *reinterpret_cast<const u32x4*>(bw + ((nb * 9 + tap) * 2 + 0) * 1024);
                wf[m & 1][1] = *reinterpret_cast<const u32x4*>(bw + ((nb * 9 + tap) * 2 + 1) * 1024);
            };
            load_x(0);
            load_w(0);
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                const int tap = order[m >> 1], ky = tap / 3, kx = tap - 3 * ky, nb = m & 1;
                const int sft = (ky >> 1) * 2 + (kx >> 1), par = (ky & 1) * 2 + (kx & 1);
                if (m + 1 < 18) load_w(m + 1);
                if (m == 4) load_x(1);
                if (m == 8) load_x(2);
                if (m == 12) load_x(3);
                const bf16x8 xh = __builtin_bit_cast(bf16x8, xs[sft & 1][0]), xl = __builtin_bit_cast(bf16x8, xs[sft & 1][1]);
                const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[m & 1][0]), wlo = __builtin_bit_cast(bf16x8, wf[m & 1][1]);
                acc[par][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc[par][nb], 0, 0, 0);
                acc[par][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh, acc[par][nb], 0, 0, 0);
                acc[par][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[par][nb], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m + 1 < 18 && k < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if ((m == 4 || m == 8 || m == 12) && k == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (m < 8 && k == 0) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        plane_step(std::integral_constant<int, 0>{});
        plane_step(std::integral_constant<int, 1>{});
        plane_step(std::integral_constant<int, 2>{});
        plane_step(std::integral_constant<int, 3>{});
        // (Tried: the last plane's taps parity by parity and the finished parities' projection steps inside the MFMA stream of its
        // remaining taps -- the schedule came out as written, the time did not change: the phase is VALU-bound, 32 instructions per
        // 3 MFMAs, and stays so wherever it runs.)
#pragma unroll
        for (int pq = 0; pq < 4; ++pq)
#pragma unroll
            for (int kidx = 0; kidx < 4; ++kidx) proj_step(pq, kidx);

        // ---- the nine-tap sum over the 16 x 64 output tile, two passes of nine record rows (rows 0..8, then rows 7..15)
        // the thread index passes through an opaque copy per tile: otherwise every lane-constant address of this phase is hoisted
        // out of the tile loop, stays live across the plane steps and spills
        int tg = tid;
        asm volatile("" : "+v"(tg));
        const int rg = tg & 31, hg = (tg >> 5) & 1;
        const int y0 = cur.y0, x0 = cur.x0, b = cur.b;
        const bool in_img = y0 + wave < H && x0 + rg < W;      // records of pixels outside the image contribute nothing
        const int Ho = 2 * H, Wo = 2 * W, oy0 = 2 * y0, ox0 = 2 * x0;
        const size_t simg = (size_t)a.B * Ho * Wo * 4;
        const float hi_c = a.gather_act == A_CLIP01 ? 1.f : INFINITY;
        const float lo_c = a.gather_act == A_NONE ? -INFINITY : 0.f;
        auto put_row = [&](int py, int rslot) {      // this wave's output row 2 wave + py into record row rslot
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float* rec = PT + ((rslot * 64 + 2 * rg + px) * TL::PTS + 4 * hg);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (8 * g + j < 23 || (8 * g + j < 27 && hg == 0)) rec[8 * g + j] = in_img ? pk[2 * py + px][4 * g + j] : 0.f;
            }
        };
        // Interior pixel (all nine taps inside the tile): bias + taps from constant offsets, clipped, stored to the image.
        // Record row of tile row sy is sy - base.
        auto finish_inner = [&](int ly, int lx, int base) {
            const float* p0 = PT + ((ly - 1 - base) * 64 + lx - 1) * TL::PTS;
            float o[3];
#pragma unroll
            for (int co = 0; co < 3; ++co) o[co] = gb[co];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int co = 0; co < 3; ++co) o[co] += p0[(ky * 64 + kx) * TL::PTS + (ky * 3 + kx) * 3 + co];
            const int gy = oy0 + ly, gx = ox0 + lx;
            if (gy >= Ho || gx >= Wo) return;
            float* op = a.img + (((size_t)b * Ho + gy) * Wo + gx) * a.gather_cs;
#pragma unroll
            for (int co = 0; co < 3; ++co)
                if (co < a.gather_cout) {
                    const float v = fminf(fmaxf(o[co], lo_c), hi_c);
                    if (DBG & 32) asm volatile("" ::"v"(v));      // ablation: no image stores
                    else op[co] = v;
                }
        };
        // Pixel of the tile's outermost ring or of the ring around the tile (ly = -1 / 16, lx = -1 / 64): the partial sum over THIS
        // tile's records goes to S[k], k = 2 (vertical neighbour's pixel) + (horizontal neighbour's pixel); gather_fix_kernel adds.
        auto finish_edge = [&](int ly, int lx, int base) {
            const bool rin = (unsigned)ly < 16u, cin = (unsigned)lx < 64u;
            float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sy = ly + ky - 1, sx = lx + kx - 1;
                    const bool ok = (unsigned)sy < 16u && (unsigned)sx < 64u;
                    // a tap outside the tile reads the zero record (32 floats in front of the records)
                    const float* pp = PT + (ok ? ((sy - base) * 64 + sx) * TL::PTS + (ky * 3 + kx) * 3 : -32);
#pragma unroll
                    for (int co = 0; co < 3; ++co) o[co] += pp[co];
                }
            const int gy = oy0 + ly, gx = ox0 + lx;
            if (gy < 0 || gx < 0 || gy >= Ho || gx >= Wo) return;
            const size_t pix = ((size_t)b * Ho + gy) * Wo + gx;
            const int k = (rin ? 0 : 2) + (cin ? 0 : 1);
            *reinterpret_cast<f32x4*>(a.S + k * simg + pix * 4) = f32x4{o[0], o[1], o[2], 0.f};
        };
        // one pass: tile rows 8 ps .. 8 ps + 7 and the ring pixels beside / beyond them; 512 interior candidates, 160 edge pixels
        auto pass = [&](int ps) {
            const int base = ps ? 7 : 0;
            {
                const int ly = 8 * ps + (tg >> 6), lx = tg & 63;
                if (ly > 0 && ly < 15 && lx > 0 && lx < 63) finish_inner(ly, lx, base);
            }
            if (tg < 160) {
                int ly, lx;
                if (tg < 66) ly = ps ? 16 : -1, lx = tg - 1;                        // the ring row above / below the tile
                else if (tg < 130) ly = ps ? 15 : 0, lx = tg - 66;                  // the tile's first / last row
                else if (tg < 144) ly = (ps ? 8 : 1) + ((tg - 130) >> 1), lx = (tg & 1) ? 63 : 0;     // its first / last column
                else ly = 8 * ps + ((tg - 144) >> 1), lx = (tg & 1) ? 64 : -1;      // the ring columns beside these eight rows
                finish_edge(ly, lx, base);
            }
        };
        auto bar = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS traffic is done (and no access moves across)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (DBG & 1) {
            asm volatile("" ::"v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]));
            if (!have_next) break;
            t = tn;
            cur = nxt;
            continue;
        }
        bar();      // every wave is done with plane 3's buffers (A1, W1): the records may overwrite them
        if (DBG & 16) {
        } else if (wave < 4) {
            put_row(0, 2 * wave);
            put_row(1, 2 * wave + 1);
        } else if (wave == 4) {
            put_row(0, 8);
        }
        bar();
        if (!(DBG & 8)) pass(0);
        bar();
        if (DBG & 16) {
            asm volatile("" ::"v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]));
        } else if (wave == 3) {
            put_row(1, 0);                                    // row 7
        } else if (wave >= 4) {
            put_row(0, 2 * wave - 7);
            put_row(1, 2 * wave - 6);
        }
        bar();
        if (!(DBG & 8)) pass(1);
        if (!have_next) break;
        t = tn;
        cur = nxt;
        // the next tile's first plane step starts with a barrier: no wave issues DMA into A1 / W1 before every wave has read its records
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int tail_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int DBG>
static hipError_t launch_tail_t(const TailArgs& a, int grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, TL::LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((tail_kernel<DBG>), dim3(grid), dim3(512), TL::LDS, s, a);
    return hipGetLastError();
}

hipError_t launch_tail(TailArgs a, hipStream_t s) {
    if (a.B < 1 || a.H < 1 || a.W < 1 || !a.in || !a.wpack || !a.bias || !a.proj_w || !a.gbias || !a.img || !a.S) return hipErrorInvalidValue;
    if (a.gather_cout < 1 || a.gather_cout > 3 || a.gather_cs < a.gather_cout) return hipErrorInvalidValue;
    if (a.gather_act != A_NONE && a.gather_act != A_RELU && a.gather_act != A_CLIP01) return hipErrorInvalidValue;
    if (a.act != A_RELU) return hipErrorInvalidValue;      // the only activation between the two layers (reference layers.py:74)
    // 32-bit offsets: the tensor and the zero bytes behind it
    if ((double)a.B * a.H * a.W * 256.0 + 4096.0 > 4294967295.0 || (double)a.zero_off + 4096.0 > 4294967295.0) return hipErrorInvalidValue;
    a.tiles_x = (a.W + TL::TWX - 1) / TL::TWX;
    a.tiles_y = (a.H + TL::TH - 1) / TL::TH;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int per_xcd = (ntiles + 7) / 8;
    int nslots = std::min(std::max(1, tail_num_cus() / 8), per_xcd);
    const int rounds = (per_xcd + nslots - 1) / nslots;
    nslots = (per_xcd + rounds - 1) / rounds;      // every workgroup the same number of tiles (+-1)
    static const int dbg = getenv("SR_TAIL_DBG") ? atoi(getenv("SR_TAIL_DBG")) : 0;
    if (dbg == 8) return launch_tail_t<8>(a, 8 * nslots, s);
    if (dbg == 16) return launch_tail_t<16>(a, 8 * nslots, s);
    if (dbg == 24) return launch_tail_t<24>(a, 8 * nslots, s);
    if (dbg == 32) return launch_tail_t<32>(a, 8 * nslots, s);
    switch (dbg & 7) {
        case 0: return launch_tail_t<0>(a, 8 * nslots, s);
        case 1: return launch_tail_t<1>(a, 8 * nslots, s);
        case 2: return launch_tail_t<2>(a, 8 * nslots, s);
        case 3: return launch_tail_t<3>(a, 8 * nslots, s);
        case 4: return launch_tail_t<4>(a, 8 * nslots, s);
        default: return launch_tail_t<7>(a, 8 * nslots, s);
    }
}

}  // namespace sr
