// The "mx6" activation format of conv_trunk6.hip -- per pixel and 32-channel block: 32 f16 hi parts, the FP6 (E2M3, one E8M0 scale)
// image of the values and the FP6 image of value - hi -- as register-level encode / decode helpers, shared with conv_trunk.hip
// (the head conv writes the format).  See conv_trunk6.hip for the layout in memory.
#pragma once
#include <hip/hip_runtime.h>

namespace sr {
namespace mx6 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

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
// element 0 whatever k is (hipcc, ROCm 7.2: found with a raw dump of the records; tests/test_gpu_mx6.py::test_format_round_trip)
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


// lane (r, h) takes all 32 channels of pixel (M block h, column r) of a finished tile: acc0 / acc1 = the accumulators of M blocks
// 0 / 1 (32 output channels x 32 pixels each, v_mfma_f32_32x32x*), V0 / V1 = lane half 0's / 1's registers of that pixel
__device__ __forceinline__ void swap_halves(const f32x16& acc0, const f32x16& acc1, f32x16& V0, f32x16& V1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index (hipcc, ROCm 7.2)
        const float a0 = acc0[i], a1 = acc1[i];
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1), false, false);
        const unsigned s0 = sw[0], s1 = sw[1];
        V0[i] = __builtin_bit_cast(float, s0);
        V1[i] = __builtin_bit_cast(float, s1);
    }
}

}  // namespace mx6
}  // namespace sr
