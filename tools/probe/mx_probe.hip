// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (E2M3) operands on gfx950: operand packing, scale semantics, throughput,
// and the f32 -> fp6 conversion builtins.  Build: hipcc --offload-arch=gfx950 -O3 -o mx_probe mx_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

// e2m3 code (0..31 magnitude) -> value
static float fp6_val(int code) {
    const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
    const float a = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -a : a;
}

// a[lane][6 dwords], b[lane][6 dwords]: 32 fp6 per lane, element e at bits [6e, 6e+6) of the lane's 192-bit string (assumption)
__global__ void mx_kernel(const unsigned* a, const unsigned* b, float* d, int sa, int sb) {
    const int lane = threadIdx.x;
    v8i va = {0, 0, 0, 0, 0, 0, 0, 0}, vb = va;
    for (int i = 0; i < 6; ++i) {
        va[i] = (int)a[lane * 6 + i];
        vb[i] = (int)b[lane * 6 + i];
    }
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 2, 2, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) d[lane * 16 + i] = c[i];
}

__global__ void time_kernel(unsigned long long* out, int iters, int which) {
    v8i va = {0x11111111, 0x22222222, 0x33333333, 0x44444444, 0x55555555, 0x66666666, 0, 0}, vb = va;
    v8h ha, hb;
    for (int i = 0; i < 8; ++i) ha[i] = hb[i] = (_Float16)(0.01f * (threadIdx.x + i));
    v16f c[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) c[k][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (which == 0) c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 2, 2, 0, 127, 0, 127);
            else if (which == 1) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[k], 0, 0, 0);
            else c[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c[k], 0, 0, 0, 127, 0, 127);   // fp8 e4m3
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += c[k][i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[which] = t1 - t0;
    if (s == 12345.678f) out[7] = 1;
}

// f32 -> fp6 conversion builtins: what lands where
__global__ void cvt_kernel(const float* in, unsigned* out, float scale) {
    v16f x0, x1;
    for (int i = 0; i < 16; ++i) {
        x0[i] = in[i];
        x1[i] = in[16 + i];
    }
    v6i r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x0, x1, scale);
    for (int i = 0; i < 6; ++i) out[i] = (unsigned)r[i];
    v6i r2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_convertvector(__builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
                                                                                                    23, 24, 25, 26, 27, 28, 29, 30, 31),
                                                                              _Float16 __attribute__((ext_vector_type(32)))),
                                                        scale);
    for (int i = 0; i < 6; ++i) out[6 + i] = (unsigned)r2[i];
}

static void put6(unsigned* w, int e, int code) {       // element e of a lane's 32
    const int bit = 6 * e;
    w[bit >> 5] |= (unsigned)code << (bit & 31);
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= (unsigned)code >> (32 - (bit & 31));
}

int main() {
    unsigned *da, *db;
    float* dd;
    hipMalloc(&da, 64 * 6 * 4);
    hipMalloc(&db, 64 * 6 * 4);
    hipMalloc(&dd, 64 * 16 * 4);
    std::vector<unsigned> a(64 * 6), b(64 * 6);
    std::vector<float> d(64 * 16);
    // reference with the assumed maps: lane l = (r = l & 31, h = l >> 5) holds k = 32 h + e of row / column r;
    // D lane (j = l & 31, hh = l >> 5) register i: row (i & 3) + 8 (i >> 2) + 4 hh, column j
    srand(1);
    std::vector<int> ca(32 * 64), cb(64 * 32);
    for (auto& v : ca) v = rand() & 63;
    for (auto& v : cb) v = rand() & 63;
    std::fill(a.begin(), a.end(), 0u);
    std::fill(b.begin(), b.end(), 0u);
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 32; ++e) {
            put6(&a[l * 6], e, ca[(l & 31) * 64 + 32 * (l >> 5) + e]);
            put6(&b[l * 6], e, cb[(32 * (l >> 5) + e) * 32 + (l & 31)]);
        }
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    for (int trial = 0; trial < 3; ++trial) {
        const int sa = trial == 0 ? 127 : (trial == 1 ? 129 : 127), sb = trial == 2 ? 124 : 127;
        hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
        hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        int bad = 0;
        const double sc = std::ldexp(1.0, sa - 127) * std::ldexp(1.0, sb - 127);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)fp6_val(ca[row * 64 + k]) * fp6_val(cb[k * 32 + col]);
                ref *= sc;
                const double err = std::fabs(ref - d[l * 16 + i]);
                if (err > 1e-3 * (1 + std::fabs(ref))) ++bad;
                if (err > worst) worst = err;
            }
        printf("mx fp6 32x32x64 scale_a %d scale_b %d: %d of 1024 outputs off the assumed layout, worst abs err %.4g (d[0] = %g)\n", sa, sb, bad, worst, d[0]);
    }
    unsigned long long* dt;
    hipMalloc(&dt, 64);
    hipMemset(dt, 0, 64);
    for (int which = 0; which < 3; ++which) hipLaunchKernelGGL(time_kernel, dim3(1), dim3(64), 0, 0, dt, 2000, which);
    unsigned long long t[8];
    hipMemcpy(t, dt, 64, hipMemcpyDeviceToHost);
    printf("cycles per MFMA, one wave, 4 independent accumulators: fp6 32x32x64 %.1f, f16 32x32x16 %.1f, fp8 32x32x64 %.1f\n", t[0] / 8000.0, t[1] / 8000.0,
           t[2] / 8000.0);
    // conversions
    float hin[32];
    for (int i = 0; i < 32; ++i) hin[i] = (i < 16 ? 0.125f * i : -0.25f * (i - 15));   // 0, .125, ... 1.875 | -0.25 ... -4
    float* din;
    unsigned* dout;
    hipMalloc(&din, 128);
    hipMalloc(&dout, 64);
    hipMemcpy(din, hin, 128, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f}) {
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(1), 0, 0, din, dout, scale);
        unsigned o[12];
        hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
        printf("cvt scale %g: 2xpk16_f32 ->", scale);
        for (int e = 0; e < 32; ++e) {
            const int bit = 6 * e;
            unsigned long long w = o[bit >> 5] | ((unsigned long long)((bit >> 5) + 1 < 6 ? o[(bit >> 5) + 1] : 0) << 32);
            printf(" %g", fp6_val((int)((w >> (bit & 31)) & 63)));
        }
        printf("\n               pk32_f16   ->");
        for (int e = 0; e < 32; ++e) {
            const int bit = 6 * e;
            unsigned long long w = o[6 + (bit >> 5)] | ((unsigned long long)((bit >> 5) + 1 < 6 ? o[6 + (bit >> 5) + 1] : 0) << 32);
            printf(" %g", fp6_val((int)((w >> (bit & 31)) & 63)));
        }
        printf("\n");
    }
    hipDeviceSynchronize();
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
