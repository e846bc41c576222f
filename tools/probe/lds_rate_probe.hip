// LDS read rate of one CU (gfx950): 8 waves, conflict-free ds_read_b128 / b64 / b32, whole chip, HIP events + s_memrealtime-free.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_rate_probe lds_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int BYTES, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void lds_kernel(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    const unsigned char* base = lds + lane * BYTES + wave * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned char* p = base + ((k * 1024 + (it & 3) * 16384) & 65535 & ~(BYTES - 1));
            if constexpr (BYTES == 16) {
                u32x4 v = *reinterpret_cast<const volatile u32x4*>(p);
                acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            } else if constexpr (BYTES == 8) {
                u32x2 v = *reinterpret_cast<const volatile u32x2*>(p);
                acc ^= v[0] ^ v[1];
            } else {
                acc ^= *reinterpret_cast<const volatile unsigned*>(p);
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES, int WAVES>
static void run(const char* name, unsigned* d) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((lds_kernel<BYTES, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, 1000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((lds_kernel<BYTES, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)iters * 16 * WAVES * 64 * BYTES;
    printf("%-28s %8.3f ms  %7.1f GB/s per CU = %.1f B/clk at 2.0 GHz (%.1f at 2.4 GHz)\n", name, ms, bytes_per_cu / ms * 1e-6, bytes_per_cu / ms * 1e-6 / 2.0,
           bytes_per_cu / ms * 1e-6 / 2.4);
}

int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    run<16, 8>("ds_read_b128, 8 waves / CU", d);
    run<16, 16>("ds_read_b128, 16 waves / CU", d);
    run<8, 8>("ds_read_b64, 8 waves / CU", d);
    run<8, 16>("ds_read_b64, 16 waves / CU", d);
    run<4, 16>("ds_read_b32, 16 waves / CU", d);
    run<16, 8>("ds_read_b128, 8 waves again", d);
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
