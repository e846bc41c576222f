// Host-side internals shared by sr_api.hip (inference step, C ABI) and sr_train.hip (training step).
#pragma once
#include "../../include/sr_hip.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct sr_train_state;   // sr_train.hip
struct sr_vgg;           // sr_vgg.h

namespace srh {

struct HostTensor {
    std::string name;
    std::vector<int64_t> shape;
    std::vector<float> data;
};

// One element of a packed weight buffer that is not a padding zero: packed[dst] = kernel[src], where
// src indexes the layer's Keras-layout kernel and dst counts elements (float, or bf16 with the low
// half of a bf16x3 split 512 elements = 64 vectors further on).  The same list drives the host packer
// (sr_commit_weights) and the device re-pack after an optimizer step (sr_train.hip).
struct PackEnt {
    int32_t src;
    uint32_t dst;
};

enum PackKind { PK_FWD = 0, PK_DGRAD = 1 };

struct ConvLayer {
    int mode = sr::M_CONV;
    int cin = 0, cout = 0, cin_p = 0;
    int nt = 1, ncb = 1, steps = 1, nchunk = 1, wstride_cb = 0;
    bool tapsn = false;   // <= 3 output channels: taps folded into N (conv_tapsn.hip)
    void* d_w = nullptr;
    float* d_bias = nullptr;
    void* d_proj = nullptr;   // transposed conv fused with the following <= 3-channel conv: its tap-projection weights
    // second packing of the same kernel for the split-planar trunk (conv_trunk.hip): SRNet's 21 trunk convs in
    // trunk_entries order; conv2d_transpose_1 in its usual order but with the k index <-> channel permutation of a
    // split-planar input (kernels.h split_perm)
    void* d_w2 = nullptr;
    std::vector<PackEnt> ents2;
    PackEnt* d_ents2 = nullptr;
    // third packing, conv2d_transpose_2 only: conv_trunk.hip's transposed-conv order for conv_tail.hip's tail_kernel
    void* d_w3 = nullptr;
    std::vector<PackEnt> ents3;
    PackEnt* d_ents3 = nullptr;
    void* d_w6 = nullptr;   // SR_F16MX6, SRNet convs 1..20: trunk6_pack_weights (conv_trunk6.hip)
    double wbytes = 0;
    // packing lists (kept for the device re-pack)
    std::vector<PackEnt> ents, proj_ents;
    PackEnt *d_ents = nullptr, *d_proj_ents = nullptr;
};

struct ProfEntry {
    std::string name;
    double ms = 0, flops = 0, bytes = 0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t round_up_sz(size_t v, size_t m = 256) { return (v + m - 1) / m * m; }

}  // namespace srh

struct sr_ctx {
    int device = 0, h = 0, w = 0, max_batch = 0, prec = sr::P_FP32, model = SR_FRVSR;
    // pad mode (sr_create_padded): the caller's tensors are (h_in, w_in); the kernels run on (h, w) = both rounded
    // up to multiples of 8, zero padded at the bottom / right, outputs cropped.  h_in == h, w_in == w otherwise.
    int h_in = 0, w_in = 0;
    bool pad = false;
    float *d_io_lr = nullptr, *d_io_hr = nullptr;   // dense caller-shaped staging for host <-> device copies (pad mode)
    // uint8 output path (sr_clip_push_u8): two device frames, converted on `stream`, copied out on `copy_stream`
    hipStream_t copy_stream = nullptr;
    unsigned char* d_u8[2] = {nullptr, nullptr};
    hipEvent_t ev_u8_ready[2] = {nullptr, nullptr}, ev_u8_done[2] = {nullptr, nullptr};
    int64_t u8_pushes = 0;
    // ... and its HOST inputs: the LR frames of push n go to d_in[n & 1] on `in_stream` when the call is made, i.e. under the
    // kernels of push n - 1 (on `stream` the copy would wait for them: 0.2 ms per step in the open)
    float* est_copy_dst = nullptr;       // sr_clip_read_est_begin's copy, not submitted yet (the next push / _end submits it)
    const float* est_copy_src = nullptr;
    size_t est_copy_bytes = 0;
    uint8_t* u8_pending_host = nullptr;  // the newest frame's destination: its D2H is submitted by the next push / sr_u8_wait(ctx, 0)
    int u8_pending_slot = 0;
    size_t u8_pending_bytes = 0;
    hipStream_t in_stream = nullptr;
    float* d_in[2] = {nullptr, nullptr};
    hipEvent_t ev_in_ready[2] = {nullptr, nullptr}, ev_in_free[2] = {nullptr, nullptr};
    bool u8_ready = false;   // every object of the path exists (set last; a failed first use frees what it made)
    // float read-back of the newest estimate beside the following step (sr_clip_read_est_begin / _end): its own stream; est_guard
    // counts the pushes that may still start before the copy has to be over (the second one overwrites the buffer it reads)
    hipStream_t est_stream = nullptr;
    hipEvent_t ev_est_src = nullptr, ev_est_done = nullptr;
    int est_guard = 0;
    bool est_pending = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // weights
    std::vector<srh::HostTensor> tensors;      // file order
    int n_fnet = 0, n_srnet = 0;
    bool committed = false;
    srh::ConvLayer fnet[14], srnet[24];
    // activations
    std::map<std::string, void*> buf;
    float *d_lr_t = nullptr, *d_lr_tm1 = nullptr, *d_est[2] = {nullptr, nullptr}, *d_flow = nullptr;
    float *d_lr_warp = nullptr, *d_hr_warp = nullptr, *d_ss = nullptr;
    // recurrent state
    bool clip_active = false;
    int clip_B = 0, cur = 0;
    int trunk_sub = 0;   // clips per trunk sub-batch (0 = whole batch)
    int planar_trunk = 1;   // SRNet trunk activations (sx, sy) in 64-byte planes; SR_NO_PLANAR=1 disables
    // bf16x3 only: trunk activations (s_in, sx, sy) split-planar, trunk convs by conv_trunk.hip.  1 = for steps of more than
    // 65,536 pixels (small steps are launch bound: the resident-weight kernel reloads 72 KiB per workgroup per launch),
    // 2 = always (SR_SPLIT_TRUNK=2, tests), 0 = never (SR_SPLIT_TRUNK=0)
    int split_trunk = 1;
    int trunk_up1 = 1;                   // split-planar path: conv2d_transpose_1 on conv_trunk.hip (SR_NO_TRUNK_UP1=1: generic kernel)
    int tail_kernel = 1;                 // ... as conv_tail.hip's plane-major tail_kernel (SR_TAIL=0: convT3x3_gather_kernel; needs trunk_up1 and fuse_gather)
    int fuse_gather = 1;                 // SRNet tail: conv2d_36's tap sum inside conv2d_transpose_2's kernel (SR_NO_FUSE_GATHER=1: tap_gather_kernel)
    int fuse_pool = 1;                   // FNet: MaxPooling2D in the producing conv's epilogue (SR_NO_FUSE_POOL=1: separate kernel)
    bool step_split = false;   // the step being enqueued uses it
    // dtype SR_F16MX6 (or SR_MX6=1 on a bf16x3 ctx): the residual blocks of a split-planar step run on conv_trunk6.hip.
    // mx6_valid: the d_w6 packings match the weights (sr_commit_weights sets it, a training update clears it)
    int trunk_mx6 = 0;
    bool mx6_valid = false;
    // bf16x3 only: FNet on conv_fnet.hip with split-planar activations: 1 = steps of more than 65,536 pixels, 2 = always
    // (SR_FNET=2, tests), 0 = never (SR_FNET=0: conv_mfma.hip)
    int fnet_split = 1;
    bool step_fnet_split = false;
    // two lanes: a step's batch is split into halves that run on `stream` and on `stream2` between a
    // fork and a join event, so one half's kernels fill the partial last round of workgroups (and the
    // under-filled low-resolution FNet launches) of the other.  SR_NO_DUAL=1 disables.
    int dual = 1;                        // number of extra lanes (0 = single stream); SR_LANES=n sets n-1
    hipStream_t stream2[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    hipStream_t lane_stream = nullptr;   // stream of the lane being enqueued (nullptr: `stream`)
    int lane_b0 = 0;                     // first image of that lane
    std::map<std::string, size_t> buf_img_bytes;   // bytes per image of every named activation buffer
    // profiling
    bool prof_on = false;
    std::vector<srh::ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    // training (sr_train.hip); the host copies in `tensors` are stale while train->host_stale
    sr_train_state* train = nullptr;
    sr_vgg* vgg = nullptr;   // VGG19 feature terms of the combined step (sr_tecogan_set_vgg); not owned
};

namespace srh {

int fail(sr_ctx* c, int code, const char* fmt, ...);

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return srh::fail(c, SR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCHCHK(c, name, call)                                                               \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return srh::fail(c, SR_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// weights
long conv_tiles(int B, int H, int W);
void plan_layer(ConvLayer& L, int prec, int mode, int cin, int cout, long tiles = 0);
std::vector<PackEnt> pack_entries(const ConvLayer& L, int prec, int kind, bool kperm = false);
std::vector<PackEnt> trunk_entries(int cin, int cout);
std::vector<PackEnt> trunkT_entries(int cin, int cout);
std::vector<PackEnt> fconv_entries(int cin, int cout);
int upload_alt(sr_ctx* c, ConvLayer& L, size_t bytes, const float* w);   // L.ents2 -> L.d_w2 (bf16x3)
std::vector<PackEnt> proj_entries(int prec, int cout);
size_t proj_bytes(int prec);
int upload_layer(sr_ctx* c, ConvLayer& L, int prec, const float* w, const float* bias, int kind = PK_FWD);
int upload_proj(sr_ctx* c, ConvLayer& L, int prec, const float* w, int cout);
void free_layer(ConvLayer& L);
extern const int FNET_SPEC[14][2];
void srnet_spec(int i, int& cin, int& cout, int& mode);

// profiling
ProfEntry* prof_entry(sr_ctx* c, const char* name);
hipEvent_t get_event(sr_ctx* c);

struct ProfScope {
    sr_ctx* c;
    size_t idx = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(sr_ctx* c_, const char* name, double flops, double bytes) : c(c_) {
        if (!c || !c->prof_on) return;
        ProfEntry* p = prof_entry(c, name);
        idx = p - c->prof.data();
        p->flops += flops;   // totals: launches sharing a name may differ in size; sr_profile_read reports the mean
        p->bytes += bytes;
        p->launches++;
        e0 = get_event(c);
        e1 = get_event(c);
        (void)hipEventRecord(e0, c->lane_stream ? c->lane_stream : c->stream);
    }
    ~ProfScope() {
        if (!e0) return;
        (void)hipEventRecord(e1, c->lane_stream ? c->lane_stream : c->stream);
        c->prof[idx].pending.emplace_back(e0, e1);
    }
};

// layer launch
struct ConvIO {
    const void* in;
    int H, W;              // input spatial size
    void* out;
    int out_cs, cout_store, out_f32;
    const void* res;
    int res_cs;
    int act;
    int in_planar = 0, out_planar = 0;   // kernels.h: planes of 64 B per pixel
    const void* proj_w = nullptr;        // transposed conv: fused tap projection, out = P [B,2H,2W,32] fp32
    int in_split = 0;                    // transposed conv, bf16x3: planar input is split-planar; uses the layer's d_w2
    int res_mask = 0;                    // kernels.h ConvArgs::res_mask: res is the activation output to mask by
    void* pool_out = nullptr;            // kernels.h ConvArgs::pool_out: store the 2x2 max pooling of the output here instead
    // kernels.h ConvArgs::gather_out: with proj_w, also gather the taps in the kernel; out = the partial-sum images, and a
    // launch_gather_fix follows inside run_conv
    float* gather_out = nullptr;
    const float* gather_bias = nullptr;
    int gather_cs = 0, gather_cout = 0, gather_act = 0;
};

int run_conv(sr_ctx* c, int prec, hipStream_t st, const ConvLayer& L, int B, const ConvIO& io, const char* name);
int check_ready(sr_ctx* c, int B);
void free_u8_path(sr_ctx* c);
// one generator step on device pointers (sr_api.hip); d_est_out / d_lr_warp / d_hr_warp may be nullptr
int run_step(sr_ctx* c, int B, const float* d_lr_t, const float* d_lr_tm1, const float* d_est_prev, float* d_est_out,
             float* d_lr_warp, float* d_hr_warp, bool warp_only);

// sr_rccl.hip: in-place sum over the ranks of `comm` (an ncclComm_t), enqueued on st
int rccl_allreduce_f32(sr_ctx* c, void* comm, float* d_buf, size_t n, hipStream_t st);

// sr_train.hip
void train_free(sr_ctx* c);
int train_sync_host(sr_ctx* c);   // refresh ctx->tensors from the device master weights after training
int train_reload(sr_ctx* c);      // after sr_commit_weights: master weights and packed buffers follow the host tensors

}  // namespace srh
