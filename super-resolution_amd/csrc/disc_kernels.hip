// Bandwidth-bound kernels of TecoGAN's spatio-temporal discriminator (reference layers.py:142-186) and of the
// adversarial losses (reference TecoGAN.py:122-131): everything that is not a convolution.  The convolutions run on the
// generator's MFMA kernels (conv_mfma.hip, train_kernels.hip): the 4x4 stride-2 'same' convolutions as 3x3 stride-1
// convolutions over the space-to-depth (block 2) image of their input, see sr_disc.hip.
//   concat9          Concatenate(axis=-1) of the nine HR images (layers.py:165-167), zero padded to 32 channels
//   d2s2             inverse of space_to_depth(2): the data gradient's way back
//   bn_*             BatchNormalization (epsilon 1e-3, momentum 0.99): batch statistics, normalise + LeakyReLU(0.2), backward
//   dense_*          Flatten + Dense(1) (layers.py:174-175) and its gradients
//   bce              sigmoid + Keras binary_crossentropy on the logits, loss and d loss / d logit
//   sqdiff_grad      mean_squared_error of a feature map / image against a target, gradient accumulated
//   teco_est_grad    gradient of the weighted losses with respect to the generator's pre-activation output
#include "disc_kernels.h"

namespace sr {

static inline int blocks_for(long n, int bs) { return (int)((n + bs - 1) / bs); }

// ------------------------------------------------------------------ concat9
__global__ void concat9_kernel(Ptr9 p, float* __restrict__ out, long npix) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * 8) return;
    const long px = idx >> 3;
    const int q = idx & 7;                       // four channels per thread
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * q + j;
        v[j] = c < 27 ? p.p[c / 3][px * 3 + c % 3] : 0.f;
    }
    reinterpret_cast<float4*>(out)[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

hipError_t launch_concat9(const Ptr9& p, float* out, long npix, hipStream_t s) {
    hipLaunchKernelGGL(concat9_kernel, dim3(blocks_for(npix * 8, 256)), dim3(256), 0, s, p, out, npix);
    return hipGetLastError();
}

// ------------------------------------------------------------------ depth_to_space(2): in [B,h,w,4C] -> out [B,2h,2w,C]
__global__ void d2s2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h, int w, int C4) {
    // one thread per 4 output channels; out[b, 2y+dy, 2x+dx, c] = in[b, y, x, (dy*2+dx)*C + c]
    const int C = C4 * 4;
    const long total = (long)B * 4 * h * w * C4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = idx % C4;
    long p = idx / C4;
    const int X = p % (2 * w);
    p /= 2 * w;
    const int Y = p % (2 * h);
    const int b = p / (2 * h);
    const int y = Y >> 1, dy = Y & 1, x = X >> 1, dx = X & 1;
    const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * h + y) * w + x) * (size_t)(4 * C) + (dy * 2 + dx) * C + 4 * c4);
    reinterpret_cast<float4*>(out)[idx] = v;
}

hipError_t launch_d2s2(const float* in, float* out, int B, int h, int w, int C, hipStream_t s) {
    if (C % 4) return hipErrorInvalidValue;
    const long total = (long)B * 4 * h * w * (C / 4);
    hipLaunchKernelGGL(d2s2_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, h, w, C / 4);
    return hipGetLastError();
}

// space_to_depth(2), vectorised: in [B,2h,2w,C] -> out [B,h,w,4C]
__global__ void s2d2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h, int w, int C4) {
    const int C = C4 * 4;
    const long total = (long)B * 4 * h * w * C4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = idx % C4;
    long p = idx / C4;
    const int X = p % (2 * w);
    p /= 2 * w;
    const int Y = p % (2 * h);
    const int b = p / (2 * h);
    const int y = Y >> 1, dy = Y & 1, x = X >> 1, dx = X & 1;
    const float4 v = reinterpret_cast<const float4*>(in)[idx];
    *reinterpret_cast<float4*>(out + (((size_t)b * h + y) * w + x) * (size_t)(4 * C) + (dy * 2 + dx) * C + 4 * c4) = v;
}

hipError_t launch_s2d2(const float* in, float* out, int B, int h, int w, int C, hipStream_t s) {
    if (C % 4) return hipErrorInvalidValue;
    const long total = (long)B * 4 * h * w * (C / 4);
    hipLaunchKernelGGL(s2d2_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, B, h, w, C / 4);
    return hipGetLastError();
}

// ------------------------------------------------------------------ BatchNormalization
// Per-channel sums over the N = B*h*w rows of a [N, C] matrix, C in {64, 128, 256}: a block of 256 threads walks
// BN_ROWS rows, thread t on channel t % C and row phase t / C; double accumulation, partial[block][2][C].
constexpr int BN_ROWS = 512;

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, double* __restrict__ partial, long N, int C) {
    __shared__ double red[2][256];
    const int c = threadIdx.x % C, ph = threadIdx.x / C, nph = 256 / C;
    const long r0 = (long)blockIdx.x * BN_ROWS;
    double s = 0.0, q = 0.0;
    for (long r = r0 + ph; r < min(N, r0 + (long)BN_ROWS); r += nph) {
        const double v = z[r * C + c];
        s += v;
        q += v * v;
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (ph == 0) {
        for (int k = 1; k < nph; ++k) {
            s += red[0][k * C + c];
            q += red[1][k * C + c];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
    }
}

// Sum of the per-block partials [nblocks][2][C] for 16 channels per 256-thread workgroup: thread (channel tid % 16, part tid / 16)
// adds blocks part, part + 16, ... and the 16 parts are added in order by the part-0 threads: a fixed order (deterministic)
// without one thread walking all blocks (44 us per call at 1024 blocks).  Returns true for the threads that hold a result.
__device__ __forceinline__ bool bn_sum_partials(const double* __restrict__ partial, int nblocks, int C, int& c, double& s, double& q) {
    __shared__ double red[2][256];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    c = blockIdx.x * 16 + cl;
    s = 0.0;
    q = 0.0;
    if (c < C)
        for (int b = part; b < nblocks; b += 16) {
            s += partial[((size_t)b * 2 + 0) * C + c];
            q += partial[((size_t)b * 2 + 1) * C + c];
        }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (part != 0 || c >= C) return false;
    for (int k = 1; k < 16; ++k) {
        s += red[0][k * 16 + cl];
        q += red[1][k * 16 + cl];
    }
    return true;
}

// mean, 1/sqrt(var + eps) of the batch (biased variance, like tf.nn.moments) and, when moving != nullptr, the moving
// statistics: moving = moving * momentum + batch * (1 - momentum), the variance fed as var * n / (n - (1 + eps))
// (keras/layers/normalization.py 2.2.4)
__global__ void bn_finalize_kernel(const double* __restrict__ partial, int nblocks, long N, int C, float eps, float momentum,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ moving_mean,
                                   float* __restrict__ moving_var) {
    int c;
    double s, q;
    if (!bn_sum_partials(partial, nblocks, C, c, s, q)) return;
    const double mean = s / (double)N;
    double var = q / (double)N - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (moving_mean) {
        const double unb = var * ((double)N / ((double)N - (1.0 + (double)eps)));
        moving_mean[c] = (float)((double)moving_mean[c] * momentum + mean * (1.0 - (double)momentum));
        moving_var[c] = (float)((double)moving_var[c] * momentum + unb * (1.0 - (double)momentum));
    }
}

// inference mode: mean = moving mean, invstd from the moving variance
__global__ void bn_moving_kernel(const float* __restrict__ mm, const float* __restrict__ mv, int C, float eps,
                                 float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean_out[c] = mm[c];
    invstd_out[c] = 1.f / sqrtf(mv[c] + eps);
}

size_t bn_workspace_bytes(long N, int C) { return (size_t)blocks_for(N, BN_ROWS) * 2 * C * sizeof(double); }

hipError_t launch_bn_stats(const float* z, long N, int C, double* partial, float eps, float momentum, float* mean, float* invstd,
                           float* moving_mean, float* moving_var, hipStream_t s) {
    if (C > 256 || 256 % C) return hipErrorInvalidValue;
    const int nb = blocks_for(N, BN_ROWS);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nb), dim3(256), 0, s, z, partial, N, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(blocks_for(C, 16)), dim3(256), 0, s, partial, nb, N, C, eps, momentum, mean, invstd,
                       moving_mean, moving_var);
    return hipGetLastError();
}

hipError_t launch_bn_moving(const float* mm, const float* mv, int C, float eps, float* mean, float* invstd, hipStream_t s) {
    hipLaunchKernelGGL(bn_moving_kernel, dim3(blocks_for(C, 64)), dim3(64), 0, s, mm, mv, C, eps, mean, invstd);
    return hipGetLastError();
}

// y = LeakyReLU_0.2(gamma * (z - mean) * invstd + beta), four channels per thread
__global__ void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, long n4, int C4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const int c = (idx % C4) * 4;
    const float4 v = reinterpret_cast<const float4*>(z)[idx];
    const float in[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = gamma[c + j] * ((in[j] - mean[c + j]) * invstd[c + j]) + beta[c + j];
        o[j] = t > 0.f ? t : 0.2f * t;
    }
    reinterpret_cast<float4*>(y)[idx] = make_float4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_bn_apply(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                           long N, int C, hipStream_t s) {
    const long n4 = N * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks_for(n4, 256)), dim3(256), 0, s, z, mean, invstd, gamma, beta, y, n4, C / 4);
    return hipGetLastError();
}

// backward, pass 1: g = gy * LeakyReLU'(y); partial sums of g and g * xhat per channel
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, double* __restrict__ partial, long N, int C) {
    __shared__ double red[2][256];
    const int c = threadIdx.x % C, ph = threadIdx.x / C, nph = 256 / C;
    const long r0 = (long)blockIdx.x * BN_ROWS;
    const float mu = mean[c], is = invstd[c];
    double s = 0.0, q = 0.0;
    for (long r = r0 + ph; r < min(N, r0 + (long)BN_ROWS); r += nph) {
        const float g = gy[r * C + c] * (y[r * C + c] > 0.f ? 1.f : 0.2f);
        s += (double)g;
        q += (double)g * (double)((z[r * C + c] - mu) * is);
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (ph == 0) {
        for (int k = 1; k < nph; ++k) {
            s += red[0][k * C + c];
            q += red[1][k * C + c];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
    }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, float* __restrict__ dbeta,
                                       float* __restrict__ dgamma) {
    int c;
    double s, q;
    if (!bn_sum_partials(partial, nblocks, C, c, s, q)) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// pass 2: dz = gamma * invstd * (g - dbeta / N - xhat * dgamma / N)   (batch statistics are functions of z)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ dbeta, const float* __restrict__ dgamma, float* __restrict__ dz, long n4,
                                    int C4, float inv_n) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const int c = (idx % C4) * 4;
    const float4 a = reinterpret_cast<const float4*>(gy)[idx], b = reinterpret_cast<const float4*>(y)[idx];
    const float4 v = reinterpret_cast<const float4*>(z)[idx];
    const float g4[4] = {a.x, a.y, a.z, a.w}, y4[4] = {b.x, b.y, b.z, b.w}, z4[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float g = g4[j] * (y4[j] > 0.f ? 1.f : 0.2f);
        const float xh = (z4[j] - mean[c + j]) * invstd[c + j];
        o[j] = gamma[c + j] * invstd[c + j] * (g - dbeta[c + j] * inv_n - xh * dgamma[c + j] * inv_n);
    }
    reinterpret_cast<float4*>(dz)[idx] = make_float4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_bn_bwd(const float* gy, const float* y, const float* z, const float* mean, const float* invstd, const float* gamma,
                         double* partial, float* dbeta, float* dgamma, float* dz, long N, int C, hipStream_t s) {
    if (C > 256 || 256 % C) return hipErrorInvalidValue;
    const int nb = blocks_for(N, BN_ROWS);
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(nb), dim3(256), 0, s, gy, y, z, mean, invstd, partial, N, C);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(blocks_for(C, 16)), dim3(256), 0, s, partial, nb, C, dbeta, dgamma);
    const long n4 = N * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(n4, 256)), dim3(256), 0, s, gy, y, z, mean, invstd, gamma, dbeta, dgamma, dz,
                       n4, C / 4, (float)(1.0 / (double)N));
    return hipGetLastError();
}

// ------------------------------------------------------------------ Flatten + Dense(1)
// logit[b] = bias + sum_f y[b, f] w[f]; one workgroup per sample
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ logit, long F) {
    __shared__ double red[256];
    const float* row = y + (size_t)blockIdx.x * F;
    double s = 0.0;
    for (long f = threadIdx.x; f < F; f += 256) s += (double)row[f] * (double)w[f];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) logit[blockIdx.x] = (float)(red[0] + (double)bias[0]);
}

hipError_t launch_dense_fwd(const float* y, const float* w, const float* bias, float* logit, int B, long F, hipStream_t s) {
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(B), dim3(256), 0, s, y, w, bias, logit, F);
    return hipGetLastError();
}

// dw[f] = sum_b dlogit[b] y[b, f]; gy[b, f] = dlogit[b] w[f]; db = sum_b dlogit[b] (thread 0 of block 0)
__global__ void dense_bwd_kernel(const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ dlogit,
                                 float* __restrict__ dw, float* __restrict__ db, float* __restrict__ gy, int B, long F) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f == 0 && db) {
        double t = 0.0;
        for (int b = 0; b < B; ++b) t += dlogit[b];
        db[0] = (float)t;
    }
    if (f >= F) return;
    const float wf = w[f];
    double acc = 0.0;
    for (int b = 0; b < B; ++b) {
        const float d = dlogit[b];
        acc += (double)d * (double)y[(size_t)b * F + f];
        gy[(size_t)b * F + f] = d * wf;
    }
    if (dw) dw[f] = (float)acc;
}

hipError_t launch_dense_bwd(const float* y, const float* w, const float* dlogit, float* dw, float* db, float* gy, int B, long F,
                            hipStream_t s) {
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(blocks_for(F, 256)), dim3(256), 0, s, y, w, dlogit, dw, db, gy, B, F);
    return hipGetLastError();
}

// ------------------------------------------------------------------ sigmoid + Keras binary_crossentropy
// K.binary_crossentropy clips the probability to [1e-7, 1 - 1e-7], goes back to a logit and applies
// sigmoid_cross_entropy_with_logits; loss = mean over the batch; dlogit = scale * (p - label) / B inside the clip range
__global__ void bce_kernel(const float* __restrict__ logit, const float* __restrict__ labels, float label_const, int B, float scale,
                           float* __restrict__ prob, float* __restrict__ dlogit, float* __restrict__ loss_out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const double z = logit[b];
        const double p = 1.0 / (1.0 + exp(-z));
        const double y = labels ? (double)labels[b] : (double)label_const;
        const double eps = 1e-7;
        const double pc = fmin(fmax(p, eps), 1.0 - eps);
        const double zc = log(pc / (1.0 - pc));
        acc += fmax(zc, 0.0) - zc * y + log1p(exp(-fabs(zc)));
        if (prob) prob[b] = (float)p;
        if (dlogit) dlogit[b] = (p > eps && p < 1.0 - eps) ? (float)((double)scale * (pc - y) / (double)B) : 0.f;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_out) *loss_out = (float)(red[0] / (double)B);
}

hipError_t launch_bce(const float* logit, const float* labels, float label_const, int B, float scale, float* prob, float* dlogit,
                      float* loss_out, hipStream_t s) {
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, s, logit, labels, label_const, B, scale, prob, dlogit, loss_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ mean squared error against a target
// loss = mean (a - t)^2 over n elements; g (+)= weight * 2 (a - t) / n  (g may be nullptr; accumulate: add to g)
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ g,
                                                     double* __restrict__ partial, long n, float gscale, int accumulate) {
    __shared__ double red[256];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double sq = 0.0;
    if (i < n) {
        const float d = a[i] - t[i];
        sq = (double)d * d;
        if (g) g[i] = (accumulate ? g[i] : 0.f) + gscale * d;
    }
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sum_partials2_kernel(const double* __restrict__ partial, int n, double inv_count, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] * inv_count);
}

size_t sqdiff_workspace_bytes(long n) { return (size_t)blocks_for(n, 256) * sizeof(double); }

hipError_t launch_sqdiff_grad(const float* a, const float* t, float* g, double* partial, float* loss_out, long n, float weight,
                              int accumulate, hipStream_t s) {
    const int nb = blocks_for(n, 256);
    hipLaunchKernelGGL(sqdiff_kernel, dim3(nb), dim3(256), 0, s, a, t, g, partial, n, (float)(2.0 * (double)weight / (double)n), accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_partials2_kernel, dim3(1), dim3(256), 0, s, partial, nb, 1.0 / (double)n, loss_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ VGG19 feature terms (sr_vgg.hip)
// loss = mean (fp - ft)^2 over n elements of two post-ReLU feature maps; g (+)= gscale * (fp - ft) where fp > 0 (the gradient
// with respect to the tapped conv's PRE-activation: ReLU is inside the Keras layer, reference TecoGAN.py:872)
__global__ __launch_bounds__(256) void feat_mse_grad_kernel(const float* __restrict__ ft, const float* __restrict__ fp, float* __restrict__ g,
                                                            double* __restrict__ partial, long n, float gscale, int accumulate) {
    __shared__ double red[256];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double sq = 0.0;
    if (i < n) {
        const float y = fp[i];
        const float d = y - ft[i];
        sq = (double)d * d;
        if (g) g[i] = (accumulate ? g[i] : 0.f) + (y > 0.f ? gscale * d : 0.f);
    }
    red[threadIdx.x] = sq;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

hipError_t launch_feat_mse_grad(const float* ft, const float* fp, float* g, double* partial, float* loss_out, long n, float weight,
                                int accumulate, hipStream_t s) {
    const int nb = blocks_for(n, 256);
    hipLaunchKernelGGL(feat_mse_grad_kernel, dim3(nb), dim3(256), 0, s, ft, fp, g, partial, n, (float)(2.0 * (double)weight / (double)n), accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_partials2_kernel, dim3(1), dim3(256), 0, s, partial, nb, 1.0 / (double)n, loss_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ the 4x4 kernel's gradient out of the 3x3 one's
// dW[ky,kx,c,o] = dW3[dy+1, dx+1, (py*2+px)*C + c, o] with ky = 2 dy + py + 1 (sr_disc.hip: conv4x4 s2 = conv3x3 on space_to_depth)
__global__ void gather_w4_kernel(const float* __restrict__ dw3, float* __restrict__ dw4, int C, int O) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)16 * C * O;
    if (idx >= total) return;
    const int o = idx % O;
    long r = idx / O;
    const int c = r % C;
    r /= C;
    const int kx = r & 3, ky = r >> 2;
    const int ty = ky + 1, tx = kx + 1;                 // t = 2 (d + 1) + p
    const int tap = (ty >> 1) * 3 + (tx >> 1), ph = (ty & 1) * 2 + (tx & 1);
    dw4[idx] = dw3[((size_t)tap * 4 * C + (size_t)ph * C + c) * O + o];
}

hipError_t launch_gather_w4(const float* dw3, float* dw4, int C, int O, hipStream_t s) {
    const long total = (long)16 * C * O;
    hipLaunchKernelGGL(gather_w4_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, dw3, dw4, C, O);
    return hipGetLastError();
}

// ------------------------------------------------------------------ generator side of the combined model
// Gradient of  w_gt * mse(est, hr) + w_pp * mse(est, pp) + (what the discriminator sends back)  with respect to the
// generator's pre-activation output, est = tanh(clip01(s) + bilinear x4 (lr)) (reference TecoGAN.py:74-76): the
// discriminator sees est as its inputs 1 and 4 (channels 3..5 and 12..14 of its 27, TecoGAN.py:97-99).
// g_out [npix, 4] (the layout train_backward starts from), partial[2][blocks] = sums of the two squared errors.
__global__ __launch_bounds__(256) void teco_est_grad_kernel(const float* __restrict__ est, const float* __restrict__ hr,
                                                            const float* __restrict__ pp, const float* __restrict__ g27,
                                                            const float* __restrict__ s_out, float* __restrict__ g_out,
                                                            double* __restrict__ partial, long npix, float sc_gt, float sc_pp,
                                                            const float* __restrict__ gvgg) {
    __shared__ double red[2][256];
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double q0 = 0.0, q1 = 0.0;
    if (idx < npix * 4) {
        const long p = idx >> 2;
        const int c = idx & 3;
        float gv = 0.f;
        if (c < 3) {
            const float a = est[p * 3 + c];
            const float d0 = a - hr[p * 3 + c], d1 = a - pp[p * 3 + c];
            q0 = (double)d0 * d0;
            q1 = (double)d1 * d1;
            gv = sc_gt * d0 + sc_pp * d1;
            if (g27) gv += g27[p * 32 + 3 + c] + g27[p * 32 + 12 + c];
            if (gvgg) gv += gvgg[p * 3 + c];                         // the four VGG19 feature terms (sr_vgg.hip), already weighted
            gv *= 1.f - a * a;                                       // tanh
            const float u = s_out[p * 3 + c];
            if (!(u > 0.f && u < 1.f)) gv = 0.f;                     // ReLU(max_value = 1) of SRNet's last layer
        }
        g_out[idx] = gv;
    }
    red[0][threadIdx.x] = q0;
    red[1][threadIdx.x] = q1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st];
            red[1][threadIdx.x] += red[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0][0];
        partial[gridDim.x + blockIdx.x] = red[1][0];
    }
}

size_t teco_est_grad_workspace_bytes(long npix) { return (size_t)2 * blocks_for(npix * 4, 256) * sizeof(double); }

hipError_t launch_teco_est_grad(const float* est, const float* hr, const float* pp, const float* g27, const float* s_out, float* g_out,
                                double* partial, float* loss_gt, float* loss_pp, long npix, float w_gt, float w_pp, hipStream_t s,
                                const float* gvgg) {
    const int nb = blocks_for(npix * 4, 256);
    const double cnt = (double)npix * 3.0;
    hipLaunchKernelGGL(teco_est_grad_kernel, dim3(nb), dim3(256), 0, s, est, hr, pp, g27, s_out, g_out, partial, npix,
                       (float)(2.0 * w_gt / cnt), (float)(2.0 * w_pp / cnt), gvgg);
    hipLaunchKernelGGL(sum_partials2_kernel, dim3(1), dim3(256), 0, s, partial, nb, 1.0 / cnt, loss_gt);
    hipLaunchKernelGGL(sum_partials2_kernel, dim3(1), dim3(256), 0, s, partial + nb, nb, 1.0 / cnt, loss_pp);
    return hipGetLastError();
}

// The discriminator's gradient with respect to its input 3, the warped previous estimate (channels 9..11), added to the
// gradient of SRNet's input, which holds the same tensor in space_to_depth order: g_sin[b,y,x, 3 + (dy*4+dx)*3 + c]
__global__ void add_hat_s2d_kernel(const float* __restrict__ g27, float* __restrict__ g_sin, int B, int H, int W, int gcs) {
    const long total = (long)B * 16 * H * W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int X = p % (4 * W);
    long q = p / (4 * W);
    const int Y = q % (4 * H);
    const int b = q / (4 * H);
    float* dst = g_sin + (((size_t)b * H + (Y >> 2)) * W + (X >> 2)) * gcs + 3 + ((Y & 3) * 4 + (X & 3)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c] += g27[p * 32 + 9 + c];
}

hipError_t launch_add_hat_s2d(const float* g27, float* g_sin, int B, int H, int W, int gcs, hipStream_t s) {
    const long total = (long)B * 16 * H * W;
    hipLaunchKernelGGL(add_hat_s2d_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, g27, g_sin, B, H, W, gcs);
    return hipGetLastError();
}

}  // namespace sr
