// fp6 -> f16 / f32 decode builtins and v_permlane32_swap on gfx950: scale direction, element order, lane movement.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));

__global__ void k(const float* in, float* out, float scale) {
    v16f x0, x1;
    for (int i = 0; i < 16; ++i) {
        x0[i] = in[i];
        x1[i] = in[16 + i];
    }
    v6i r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x0, x1, scale);
    v32h h = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(r, scale);
    v32f f = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, scale);
    if (threadIdx.x == 0)
        for (int i = 0; i < 32; ++i) {
            out[i] = (float)h[i];
            out[32 + i] = f[i];
        }
    // permlane32_swap: a = 100 + lane, b = 200 + lane
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[64 + threadIdx.x] = (float)s[0];
    out[128 + threadIdx.x] = (float)s[1];
}

int main() {
    float hin[32], hout[192];
    for (int i = 0; i < 32; ++i) hin[i] = (i < 16 ? 0.25f * i : -0.5f * (i - 15));
    float *din, *dout;
    hipMalloc(&din, 128);
    hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, 128, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 4.0f, 0.25f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, scale);
        hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        printf("scale %g\n  in      :", scale);
        for (int i = 0; i < 32; ++i) printf(" %g", hin[i]);
        printf("\n  f16 back:");
        for (int i = 0; i < 32; ++i) printf(" %g", hout[i]);
        printf("\n  f32 back:");
        for (int i = 0; i < 32; ++i) printf(" %g", hout[32 + i]);
        printf("\n");
    }
    printf("permlane32_swap(a = 100 + lane, b = 200 + lane):\n  result[0]:");
    for (int i = 0; i < 64; ++i) printf(" %g", hout[64 + i]);
    printf("\n  result[1]:");
    for (int i = 0; i < 64; ++i) printf(" %g", hout[128 + i]);
    printf("\nlast error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
