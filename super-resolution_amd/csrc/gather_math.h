// Device arithmetic shared by the forward gathers (fused_ops.hip) and their gradients
// (train_kernels.hip): the TF1-legacy bilinear lerp and dense_image_warp's coordinates.  The backward
// kernels must take the same floor / clamp decisions as the forward ones, so both use these.
#pragma once
#include <hip/hip_runtime.h>

namespace sr {

// legacy TF1 bilinear: top = tl + (tr-tl)*tx ; bot likewise ; out = top + (bot-top)*ty
__device__ __forceinline__ float lerp2(float tl, float tr, float bl, float br, float tx, float ty) {
    const float top = tl + (tr - tl) * tx;
    const float bot = bl + (br - bl) * tx;
    return top + (bot - top) * ty;
}

// ------------------------------------------------------------------ dense_image_warp helpers
struct WarpCoord {
    int f;
    float a;
};
// floor clamped to [0, n-2]; alpha from the CLAMPED floor, clamped to [0,1]  
__device__ __forceinline__ WarpCoord warp_coord(float q, int n) {
    float f = floorf(q);
    f = fminf(fmaxf(f, 0.f), (float)(n - 2));
    WarpCoord w;
    w.f = (int)f;
    w.a = fminf(fmaxf(q - f, 0.f), 1.f);
    return w;
}
__device__ __forceinline__ float warp_interp(float tl, float tr, float bl, float br, float ax, float ay) {
    const float top = ax * (tr - tl) + tl;
    const float bot = ax * (br - bl) + bl;
    return ay * (bot - top) + top;
}

}  // namespace sr
