// The one entry point of libsr_hip.so that allocates nothing (include/sr_hip.h: sr_op_conv3x3_wgrad_noalloc).
// It is sr_op_conv3x3_bwd's weight / bias gradient on DEVICE tensors for callers that cannot call hipMalloc
// (code under HIP graph capture, tight training loops, harnesses that forbid allocation).  The kernels are the
// ones sr_op_conv3x3_bwd and the training step launch (train_kernels.hip: wgrad_kernel / wgrad_x3_kernel write one
// partial sum per pixel slice, wgrad_reduce_kernel adds them in a fixed order); only the home of the slice
// partials differs: a static device array instead of a hipMalloc'ed workspace.
#include "sr_internal.h"

#include <mutex>

using namespace sr;

namespace srh {
extern thread_local std::string g_err;
}

// Slice partials of one weight-gradient launch.  Every GPU the library runs on holds its own copy of this
// array, at its own address, until the process ends.  In host code `scratch` is only a handle for
// hipGetSymbolAddress: the address kernels use is g_scratch[device], looked up once per device.
static __device__ __align__(256) unsigned char scratch[SR_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEV = 64;
unsigned char* g_scratch[MAX_DEV];   // per device: device address of that GPU's `scratch`
std::mutex g_scratch_mu;

// Called with `device` current.  First call on a device resolves that device's copy of the array.
int scratch_ptr(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return srh::fail(nullptr, SR_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", device,
                             hipGetErrorString(e));
        g_scratch[device] = (unsigned char*)p;
    }
    *out = g_scratch[device];
    return SR_OK;
}
}  // namespace

extern "C" int sr_op_conv3x3_wgrad_noalloc(int device, int dtype, int B, int H, int W, int Cin, int Cout, const float* d_x,
                                           const float* d_dy, float* d_dw, float* d_db, void* hip_stream) {
    using srh::fail;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SR_ERR_HIP, "no HIP device available; libsr_hip has no CPU fallback");
    if (device < 0 || device >= ndev || device >= MAX_DEV) return fail(nullptr, SR_ERR_ARG, "device %d out of range", device);
    if (!d_x || !d_dy || !d_dw) return fail(nullptr, SR_ERR_ARG, "NULL tensor pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(nullptr, SR_ERR_ARG, "non-positive dimension");
    if (dtype != SR_FP32 && dtype != SR_BF16X3) return fail(nullptr, SR_ERR_ARG, "backward ops need dtype fp32 or bf16x3");
    if ((double)B * H * W >= 2147483648.0) return fail(nullptr, SR_ERR_ARG, "more than 2^31 pixels");
    const size_t need = wgrad_workspace_bytes(B, H, W, Cin, Cout, 0);
    if (need > (size_t)SR_NOALLOC_SCRATCH_BYTES)
        return fail(nullptr, SR_ERR_ARG, "sr_op_conv3x3_wgrad_noalloc: this shape needs %zu bytes of slice partials, the static "
                    "scratch holds %d; use sr_op_conv3x3_bwd (which allocates) or a smaller batch", need,
                    (int)SR_NOALLOC_SCRATCH_BYTES);
    // the caller's current device is restored on every exit path (a multi-GPU caller passing device != current must not
    // find its later HIP calls redirected)
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } guard;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(nullptr, SR_ERR_HIP, "hipGetDevice failed");
    if (cur != device) {
        if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SR_ERR_HIP, "hipSetDevice failed");
        guard.prev = cur;
    }
    // first call on a device: hipGetSymbolAddress -- not legal inside a stream capture, so a capturing caller makes one
    // call per device (any valid shape) before it starts capturing; see include/sr_hip.h
    unsigned char* ptr = nullptr;
    if (int rc = scratch_ptr(device, &ptr)) return rc;
    // every slice partial the reduction reads is written by this launch's first kernel, so nothing has to be
    // cleared; two calls that overlap on one GPU (different streams) would share the partials and race
    WgradArgs wa = {d_x, d_dy, reinterpret_cast<float*>(ptr), d_dw, d_db, B, H, W, Cin, Cout, Cin, Cout, 0,
                    dtype == SR_BF16X3 ? 1 : 0};
    const hipError_t e = launch_wgrad(wa, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(nullptr, SR_ERR_HIP, "launch wgrad: %s", hipGetErrorString(e));
    return SR_OK;
}
