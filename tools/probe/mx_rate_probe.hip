// Issue rate of the gfx950 MFMA shapes the trunk could use, whole chip, timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -o mx_rate_probe mx_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

template <int WHICH>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    v8i va, vb;
    for (int i = 0; i < 8; ++i) {
        va[i] = 0x11111111 * (i + 1) + threadIdx.x;
        vb[i] = 0x01010101 * (i + 3) + threadIdx.x;
    }
    v8h ha, hb;
    v8b ba, bb;
    for (int i = 0; i < 8; ++i) {
        ha[i] = hb[i] = (_Float16)(0.01f * (threadIdx.x + i));
        ba[i] = bb[i] = (__bf16)(0.01f * (threadIdx.x + i));
    }
    v16f c[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) c[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (WHICH == 0) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[k], 0, 0, 0);
            if constexpr (WHICH == 1) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c[k], 0, 0, 0);
            if constexpr (WHICH == 2) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 0, 0, 0, 127, 0, 127);   // fp8
            if constexpr (WHICH == 3) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 2, 2, 0, 127, 0, 127);   // fp6 e2m3
            if constexpr (WHICH == 4) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 4, 4, 0, 127, 0, 127);   // fp4
            if constexpr (WHICH == 5) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 2, 2, 0, 127 + (it & 1), 0, 126);   // fp6, scales that change
            if constexpr (WHICH == 6) {   // the mix one channel block of the proposed trunk would issue: 4 f16 + 2 fp6 per 64 channels
                c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[k], 0, 0, 0);
                c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, c[k], 0, 0, 0);
                c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 2, 2, 0, 127, 0, 126);
            }
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += c[k][i];
    if (s == 12345.678f) out[0] = s;
}

template <int WHICH>
static void run(const char* name, float* d, int per_iter) {
    const int iters = 20000, blocks = 512;          // 2 blocks of 4 waves per CU: 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<WHICH>, dim3(blocks), dim3(256), 0, 0, d, 2000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<WHICH>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * per_iter * 2;      // MFMAs each SIMD issued
    printf("%-34s %8.3f ms  %7.2f ns per MFMA per SIMD (= %.1f cycles at 2.0 GHz, %.1f at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.0,
           ms * 1e6 / per_simd * 2.4);
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    run<0>("f16 32x32x16", d, 4);
    run<1>("bf16 32x32x16", d, 4);
    run<2>("mx fp8 32x32x64", d, 4);
    run<3>("mx fp6 32x32x64", d, 4);
    run<4>("mx fp4 32x32x64", d, 4);
    run<5>("mx fp6 32x32x64, varying scales", d, 4);
    run<6>("2 f16 x16 + 1 fp6 x64 (per group)", d, 12);
    run<0>("f16 32x32x16 again", d, 4);
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
